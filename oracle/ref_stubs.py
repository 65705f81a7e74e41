"""TEST INFRASTRUCTURE ONLY - never imported by the product path.

Makes the upstream reference (``/root/reference``, present only in the build
container) importable on CPU by registering stub modules for its absent,
non-arithmetic dependencies (kiui, trimesh, megfile, cv2, torchvision, tyro).
Used by ``oracle/make_golden.py`` to (a) validate the restatement in
``oracle/arae_oracle.py`` against the reference's own modules and (b) generate
the golden fixtures committed under ``tests/golden/``.

Order matters: ``transformers`` must be imported before
``torchvision`` is stubbed.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("EDGERUNNER_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "core"))


def install():
    """Register the stubs and put the reference on sys.path. Idempotent."""
    if getattr(install, "_done", False):
        return
    import torch  # noqa: F401
    import transformers  # noqa: F401
    from transformers import CLIPVisionModel, CLIPImageProcessor  # noqa: F401  (resolve lazily before stubbing)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _lo(*a, **k):
        return None

    def _seed_everything(seed, *a, **k):
        import random
        import numpy as np
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)

    kiui = mod("kiui", lo=_lo, seed_everything=_seed_everything)
    kiui.mesh_utils = mod("kiui.mesh_utils", clean_mesh=None, decimate_mesh=None)
    kiui.op = mod("kiui.op", recenter=None)
    mod("trimesh")
    mod("megfile")
    mod("cv2")
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional", normalize=None)

    def _subcommand_type_from_defaults(defaults, descriptions=None, **k):
        return dict

    tyro = mod("tyro")
    tyro.extras = mod("tyro.extras", subcommand_type_from_defaults=_subcommand_type_from_defaults)
    tyro.cli = lambda *a, **k: None

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True
