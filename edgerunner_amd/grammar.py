"""Token grammars of ``LMM.generate`` (reference: core/models.py:236-275).

The reference expresses them as a Python ``prefix_allowed_tokens_fn`` closure that
HuggingFace calls once per token on the host.  On the device path they are an
enum (``er_grammar``) evaluated inside the sampling-head kernel; this module holds
(a) the enum selection ``LMM.generate`` performs and (b) host-side equivalents used
when a caller passes an arbitrary callable (step-wise path) and by the unit tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

from . import native


def select_grammar(opt, has_tokenizer: bool) -> int:
    """Which built-in grammar ``LMM.generate`` would construct (core/models.py:236-275)."""
    if not has_tokenizer:
        return native.ER_GRAMMAR_NAIVE9
    if opt.meto_backend in ("LR", "LR_ABSCO"):
        return native.ER_GRAMMAR_LR_ABSCO
    return native.ER_GRAMMAR_NONE   # reference prints a warning and passes None


class GrammarState:
    """Host mirror of the device automaton for one sequence."""

    def __init__(self, grammar: int, vocab_size: int, eos_token_id: int = 2):
        self.grammar, self.vocab, self.eos = grammar, vocab_size, eos_token_id
        self.counter = 0
        self.t = 0          # tokens generated so far

    def allowed(self, last_token: Optional[int]) -> List[int]:
        """Allowed ids for the next token given the previously generated one
        (None before the first).  Mutates the counter like the reference closure."""
        g = self.grammar
        if g == native.ER_GRAMMAR_NONE:
            out = list(range(self.vocab))
        elif g == native.ER_GRAMMAR_NAIVE9:
            out = list(range(3, self.vocab))
            if self.t % 9 == 1:
                out.append(self.eos)
        else:
            if self.t == 0:
                out = [5]
            else:
                if last_token == 5:
                    self.counter = 9
                elif last_token in (3, 4):
                    self.counter = 3
                elif last_token is not None and last_token >= 6:
                    self.counter -= 1
                out = list(range(6, self.vocab)) if self.counter > 0 else [3, 4, 5, self.eos]
        self.t += 1
        return out


def as_callable(grammar: int, vocab_size: int, eos_token_id: int = 2) -> Optional[Callable]:
    """``prefix_allowed_tokens_fn(batch_id, ids)`` equivalent of a built-in grammar."""
    if grammar == native.ER_GRAMMAR_NONE:
        return None
    states = {}

    def fn(batch_id, ids):
        st = states.setdefault(batch_id, GrammarState(grammar, vocab_size, eos_token_id))
        last = int(ids[-1]) if len(ids) > 0 else None
        st.t = len(ids)
        return st.allowed(last)
    return fn
