#!/usr/bin/env python3
"""Benchmark of the ArAE decode hot path on MI355X (contract: see the task's bench.py section).

One "step" = one full pass of the hot path for the synthetic point clouds of one batch per GPU:
encode_cond (point encoder) -> 2050-token prefill -> T greedy tokens with the device-side
grammar head (LMM.generate; reference core/models.py:204-303).  Default workload = BASELINE.json
configs[1]: ArAE (24 layers, 1536 wide) random-init, batch 1, greedy, test_num_face=1000,
T = 4*num_faces = 4000 new tokens with EOS suppressed until T, 4096-point cloud, exact fp32
mode (the mode whose greedy ids are bit-exact vs the reference CPU path).
``--config 2`` / ``--config 3`` run the other single-GPU configurations of BASELINE.json at FULL size, each with its own
roofline block: 2 = configs[2] (batch 32, SAMPLE mode top-k 10, test_num_face=4000 -> T = 16000 tokens, context 2050 -> 18050, fp16
storage: the long-sequence KV-cache case, ~2.5 min per step), 3 = configs[3]'s per-GPU shard (32 clouds per GPU, greedy,
test_num_face=1000, T = 4000, fp16 storage).  Explicit flags (--batch-per-gpu, --num-face, --precision, --mode, --tokens) override
the configuration's values.

N > 1: one process per GPU.  ``python bench.py --gpus N`` launches the N ranks itself (re-executes under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``) unless it already runs
inside such a launch (WORLD_SIZE set, e.g. by the driver's own torchrun line).  Every rank generates for its own
clouds (weak scaling: independent samples, full weight replica per GPU, no data-path collective) and the token
streams are all-gathered once per step over RCCL.  value = total tokens / max-over-ranks time.

``--dry-run`` replaces the GPU work by fabricated token streams (gloo on CPU): it exercises the launch, sharding,
gather, barrier and max-over-ranks plumbing and prints the same JSON shape with ``"dry_run": true``.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
W_ELEMS = 680_752_128          # streamed weight elements per token
KV_ELEMS_PER_POS = 73_728      # 2 * 24 * 1536
PREFIX = 2050                  # 2049 condition tokens + BOS
LAUNCHES_PER_TOKEN = {"qkv_gemv": 24, "attn_decode": 24, "attn_combine": 24, "out_proj_gemv": 24, "fc1_gemv": 24,
                      "fc2_gemv": 24, "lm_head_gemv": 1, "sample_head": 1}
# committed rocprofv3 PMC summaries (scripts/gpu_round4.sh pmc / pmc3): single-row decode kernels, batched (B = 32) decode kernels
# committed PMC summaries, newest round first (a file that is not there yet falls through to the previous round's)
PMC_SUMMARY = {("fp32", False): ["r06_pmc_hbm_summary.json", "r05_pmc_hbm_summary.json", "r04_pmc_hbm_summary.json"],
               ("fp16", False): ["r06_pmc_hbm_fp16_summary.json", "r05_pmc_hbm_fp16_summary.json"],
               ("fp16", True): ["r06_pmc_hbm_config3_summary.json", "r05_pmc_hbm_config3_summary.json", "r04_pmc_hbm_config3_summary.json"],
               ("fp32", True): ["r06_pmc_hbm_config3_exact_summary.json"]}
# the single-GPU configurations of BASELINE.json (configs[0] is the CPU path = cpu_baseline; configs[4] = dit_front_end_fp16)
CONFIGS = {
    1: {"name": "BASELINE configs[1]", "batch": 1, "num_face": 1000, "mode": "greedy", "precision": "fp32"},
    2: {"name": "BASELINE configs[2]", "batch": 32, "num_face": 4000, "mode": "sample", "precision": "fp16"},
    3: {"name": "BASELINE configs[3] shard (32 of the 256 clouds per GPU)", "batch": 32, "num_face": 1000, "mode": "greedy",
        "precision": "fp16"},
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=(1, 2, 3), default=1,
                    help="BASELINE.json configuration: 1 = configs[1] (default: B = 1 greedy, exact fp32), 2 = configs[2] (B = 32 sample "
                         "mode, test_num_face 4000, fp16), 3 = configs[3] shard (B = 32 greedy, fp16)")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="clouds per generate() call per GPU (default: the configuration's)")
    ap.add_argument("--num-face", type=int, default=None)
    ap.add_argument("--mode", choices=("greedy", "sample"), default=None)
    ap.add_argument("--tokens", type=int, default=None, help="new tokens per sample (default 4*num_face)")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--resume-len", type=int, default=0,
                    help="extra teacher-given tokens appended to the prefix (resume_ids): starts the decode at a longer "
                         "context; used by the PMC passes to measure attention traffic at the run's mean context length")
    ap.add_argument("--cpu-steps", type=int, default=120, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=("fp32", "fp16"), default=None,
                    help="fp32 = exact mode (the mode whose ids are bit-exact vs the CPU path; configs[1]); fp16 = fast mode")
    ap.add_argument("--no-fast-extra", action="store_true", help="skip the additional fp16 fast-mode / batch-32 measurements")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: fabricated streams through the same multi-rank plumbing")
    ap.add_argument("--master-port", type=int, default=0)
    a = ap.parse_args(argv)
    cfg = CONFIGS[a.config]
    a.overridden = [k for k, v in (("batch_per_gpu", a.batch_per_gpu), ("num_face", a.num_face), ("mode", a.mode),
                                   ("precision", a.precision), ("tokens", a.tokens)) if v is not None]
    a.batch_per_gpu = cfg["batch"] if a.batch_per_gpu is None else a.batch_per_gpu
    a.num_face = cfg["num_face"] if a.num_face is None else a.num_face
    a.mode = cfg["mode"] if a.mode is None else a.mode
    a.precision = cfg["precision"] if a.precision is None else a.precision
    return a


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside a distributed launch: start the N ranks (one process per GPU)."""
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def pmc_traffic(kind, kernel_names, at_len, batched=False, precision="fp32"):
    """HBM bytes per launch of a decode kernel from the committed rocprofv3 PMC passes (profiles/r05_pmc_hbm_*summary.json:
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes; separate --pmc runs, scripts/gpu_round5.sh pmc / pmc16 / pmc3).
    Counters cannot be read inside the timed process.  The attention passes run at a recorded context length (the PMC run starts
    its decode at --resume-len); its bytes are scaled linearly to `at_len` keys, the length `bytes_per_launch` is quoted at."""
    try:
        rel = next(os.path.join("profiles", f) for f in PMC_SUMMARY[(precision, bool(batched))] if os.path.exists(os.path.join(ROOT, "profiles", f)))
        doc = json.load(open(os.path.join(ROOT, rel)))
        ks = doc["kernels"]
        def lookup(n):      # rocprofv3 leaves kernels with _Float16 template arguments mangled: match base name + element type
            if n in ks:
                return ks[n]
            base, half = n.split("<")[0], "_Float16" in n
            return next((v for m, v in ks.items() if m.startswith("_Z") and base in m and (("DF16_" in m) == half)), None)
        k = next(v for v in (lookup(n) for n in kernel_names) if v is not None)
        b = k["hbm_read_bytes_per_launch"] + k.get("hbm_write_bytes_per_launch", 0.0)
        note = ""
        if kind == "attn_decode":
            l_pmc = float(doc.get("attention_context_len_mean", 0) or 0)
            if l_pmc > 0:
                b = b * at_len / l_pmc
                note = f" (attention measured at mean context {l_pmc:.0f}, scaled x{at_len / l_pmc:.4f} to {at_len} keys)"
        return {"bytes": round(b), "source": rel + note}
    except Exception:
        return {}


class _Budget(Exception):
    pass


def cpu_baseline(opt, sd, T_sample, num_points, budget_s=45.0):
    """Reference CPU-eager path timed on this box's host cores: the oracle (a torch-CPU fp32
    restatement that is bit-identical to the reference's own modules, see oracle/) on a bounded
    sample of the same workload: encode + prefill + the first decode steps (at most T_sample
    steps or budget_s seconds of decoding, whichever comes first)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import arae_oracle as O
    from edgerunner_amd import weights as W
    default_threads = torch.get_num_threads()  # torch's default = the cores this process may use
    pc = W.synthetic_point_cloud(0, num_points)
    # pick the thread count the CPU path runs fastest with (a 1-row GEMV stream does not scale to 100+ threads):
    # time a few cached decode steps of a 2-layer slice of the same model per candidate
    cand = sorted({t for t in (8, 16, 32, 64, default_threads) if t <= default_threads})
    probe_opt = dataclasses.replace(opt, num_layers=2)
    probe_sd = {k: v for k, v in sd.items() if ".layers." not in k or int(k.split(".layers.")[1].split(".")[0]) < 2}
    emb = torch.randn(1, 2050, opt.hidden_dim) * 0.5
    timings = {}
    for t in cand:
        torch.set_num_threads(t)
        fwd = O.make_forward(probe_sd, probe_opt)
        _, past = fwd(inputs_embeds=emb)
        ids = torch.tensor([[7]])
        fwd(input_ids=ids, past=past)
        t1 = time.perf_counter()
        for _ in range(6):
            fwd(input_ids=ids, past=past)
        timings[t] = (time.perf_counter() - t1) / 6
    threads = min(timings, key=timings.get)
    torch.set_num_threads(threads)
    print(f"[bench] cpu baseline: thread probe (s/step, 2 layers) {timings} -> {threads} threads", file=sys.stderr, flush=True)
    marks = []

    def timer(t):
        marks.append(time.perf_counter())
        if len(marks) >= 3 and marks[-1] - marks[0] > budget_s:
            raise _Budget()

    t0 = time.perf_counter()
    try:
        O.lmm_generate_ids(sd, opt, pc, 1000, max_new_tokens=T_sample, min_new_tokens=T_sample, step_timer=timer)
    except _Budget:
        pass
    dec = np.diff(np.array(marks))
    n = len(dec)
    full = {}
    try:     # BASELINE.md section 2 protocol (full T, three phases): the committed run of the build container (make_golden.py full)
        m = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json")))["arae_full_T4000"]
        full = {"T": m["T"], "threads": m["cpu_threads"], "host": m.get("cpu_host", "build container"),
                "phase_i_encode_s": m.get("cpu_encode_s"), "phase_ii_prefill_s": m.get("cpu_prefill_s"),
                "phase_iii_decode_s": round(m["T"] / m["cpu_decode_tok_per_s"], 1),
                "decode_tokens_per_s": round(m["cpu_decode_tok_per_s"], 3),
                "end_to_end_tokens_per_s": round(m["T"] / m["cpu_seconds_total"], 3),
                "ms_per_token_first100": round(m["cpu_ms_per_token_first100"], 1),
                "ms_per_token_last100": round(m["cpu_ms_per_token_last100"], 1),
                "source": "tests/golden/MANIFEST.json (the 4000-token reference-module run that produced the golden ids; NOT this box)"}
    except Exception:  # noqa: BLE001
        pass
    return {
        "full_run": full,
        "value": round(float(n / dec.sum()), 3), "unit": "tokens/s", "cores": threads, "host_cpus": os.cpu_count(),
        "thread_probe_s_per_step_2layers": {str(k): round(v, 5) for k, v in timings.items()},
        "kind": "port",
        "sample": f"oracle (torch CPU fp32, = reference modules bit-for-bit): encode_cond + 2050-token prefill "
                  f"({marks[0] - t0:.1f}s) + first {n} greedy decode steps at context 2050..{2050 + n} "
                  f"({dec.sum():.1f}s); the full 4000-token run is slower per token as context grows to 6050",
        "prefill_plus_encode_s": round(marks[0] - t0, 2),
    }


def dit_front_end_extra(dev, steps=20):
    """Secondary figure (BASELINE configs[4], not `value`): the image-conditioned front-end at FULL depth on this GPU in
    the reference's GPU dtype (fp16 matrix cores): CLIP ViT-H/14 (32 layers) -> proj/norm -> DiT (24 layers, 2048 latent
    tokens) sampled with DDIM under CFG 7.5 (batch-2 forward per step).  Compute-bound: priced against the dense fp16
    MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).  FLOPs per CFG forward are algorithmic: per layer and sample
    36 C^2 N (projections + GEGLU MLP) + 4 N C (N + M) (self + cross attention), C = 1024, N = 2048, M = 257."""
    import torch
    from edgerunner_amd import weights as W
    from edgerunner_amd.models_dit import MDiT
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["DiT"], generate_mode="greedy", cond_mode="point_latent")
    m = MDiT(opt, dev, clip_layers=32, precision="fp16")
    sd = W.make_dit_state_dict(opt, 0, "perturbed")
    sd.update(W.make_clip_state_dict(32, 0, "perturbed"))
    m.load_state_dict(sd, strict=True)
    del sd
    img = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(1)).to(dev)
    out = {}
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        m.get_cond(img)
        torch.cuda.synchronize(); enc = time.perf_counter() - t
        t = time.perf_counter()
        lat = m.run(img, num_inference_steps=steps, guidance_scale=7.5)
        torch.cuda.synchronize(); tot = time.perf_counter() - t
    C, N, M, L = opt.dit_hidden_dim, opt.point_latent_size, 257, opt.dit_num_layers
    flops = 2 * L * (36.0 * C * C * N + 4.0 * N * C * (N + M))
    per_fwd = (tot - enc) / steps
    m.close()
    return {"workload": f"BASELINE configs[4] front-end, full depth: CLIP ViT-H/14 32 layers + DiT {L} layers, {steps} DDIM steps, CFG 7.5, "
                        "fp16 matrix cores, synthetic weights",
            "clip_encode_ms": round(enc * 1e3, 2), "ms_per_cfg_forward": round(per_fwd * 1e3, 3),
            "ms_per_100_steps": round(per_fwd * 1e5 + enc * 1e3, 1), "finite": bool(torch.isfinite(lat).all()),
            "roofline": {"bound": "mfma", "achieved": round(flops / per_fwd / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / per_fwd / 2.5e15, 4), "flops_per_cfg_forward": flops}}


def kernel_names(precision, batched):
    """rocprofv3 names of the decode kernels per kind for this build's default knobs (scripts/roofline_from_rocprof.py and
    the PMC summary are matched on them); gemv_kernel<WT, KS, NB, RW, PRO, EPI, NW>."""
    wt = "float" if precision == "fp32" else "_Float16"
    fast = precision != "fp32"
    nwq = os.environ.get("ER_NW_QKV", "9" if fast else "6")     # er_create's rule: 4 / 6 / 9 waves (9 = one fat workgroup per CU)
    nwq = nwq if nwq in ("4", "9") else "6"
    nwf = "12" if os.environ.get("ER_NW_FC1", "12" if fast else "4") == "12" else "4"
    qkv = f"gemv_kernel<{wt}, 1, 1, {2 if nwq == '9' else 1}, 1, 3, {nwq}>"
    fc1 = f"gemv_kernel<{wt}, 1, 1, 2, 1, 1, {nwf}>"
    rw2 = os.environ.get("ER_RW_FC2", "6") if fast else "2"       # fast mode: six rows per fc2 workgroup (er_api.hip, case 5)
    fc2 = f"gemv_kernel<{wt}, 4, 1, {rw2 if rw2 in ('4', '6') else '2'}, 0, 2, 4>"
    if batched:     # B*16 >= 256: one streaming workgroup per (row, head); smaller batches keep the split round-1 kernel (er_api.hip, kind 1)
        return {"attn_decode": [f"attn_stream_kernel<{wt}, 96, 2>", f"attn_decode_kernel<{wt}, 96, {4 if precision == 'fp32' else 2}>"]}
    if os.environ.get("ER_DECODE_V", "3") != "2":     # default: balanced chunks, merge fused into out_proj (no merge kernel)
        return {"qkv_gemv": [qkv],
                "attn_decode": [f"attn_decode3_kernel<{wt}, 96, {4 if precision == 'fp32' else 2}, 16>"],
                "out_proj_gemv": [f"outproj_merge_kernel<{wt}, 96, 16>"],
                "fc1_gemv": [fc1], "fc2_gemv": [fc2],
                "lm_head_gemv": [f"gemv_kernel<{wt}, 1, 1, 1, 1, 0, 4>"], "sample_head": ["sample_head_kernel"]}
    return {"qkv_gemv": [qkv],
            "attn_decode": [f"attn_decode2_kernel<{wt}, 96, {4 if precision == 'fp32' else 2}>"],
            "attn_combine": ["attn_combine2_kernel<96>"],
            "out_proj_gemv": [f"gemv_kernel<{wt}, 1, 1, 1, 0, 2, 3>"],
            "fc1_gemv": [fc1], "fc2_gemv": [fc2],
            "lm_head_gemv": [f"gemv_kernel<{wt}, 1, 1, 1, 1, 0, 4>"], "sample_head": ["sample_head_kernel"]}


def dry_run(args):
    """Multi-rank plumbing without a GPU: gloo, fabricated streams, same gather / barrier / max-over-ranks / JSON."""
    from edgerunner_amd import dist as D
    rank, world, _ = D.init_process_group(backend="gloo")
    B, T = args.batch_per_gpu, args.tokens or 4 * args.num_face
    n_items = world * B

    gather_ms = []

    def one_step(k):
        mine = D.shard_indices(n_items, rank, world)
        streams = [np.full(T, 6 + (i + k) % 500, dtype=np.int64) for i in mine]
        tg = time.perf_counter()
        got = D.gather_token_streams(streams, n_items)
        gather_ms.append((time.perf_counter() - tg) * 1e3)
        assert len(got) == n_items and all(int(g[0]) == 6 + (i + k) % 500 for i, g in enumerate(got))

    for w in range(args.warmup):
        one_step(-1 - w)
    gather_ms.clear()
    D.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k)
    D.barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(my_elapsed)
    rk_wall, rk_gather = D.all_ranks(my_elapsed), D.all_ranks(float(np.mean(gather_ms)) if gather_ms else 0.0)
    per_rank = {"wall_s": {"min": round(min(rk_wall), 4), "max": round(max(rk_wall), 4)},
                "gather_ms_per_step": {"min": round(min(rk_gather), 3), "max": round(max(rk_gather), 3)}, "ranks_reporting": len(rk_wall)}
    if rank == 0:
        print(json.dumps({"metric": "mesh tokens/sec (whole node), ArAE greedy test_num_face=1000", "dry_run": True,
                          "value": round(world * B * T * args.steps / max(elapsed, 1e-9), 2), "unit": "tokens/s",
                          "n_gpus": world, "world_size_seen": world, "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "data": "fabricated (dry run: no GPU work)", "per_rank": per_rank,
                          "config": {"workload": f"dry run: {B} fabricated stream(s) of {T} ids per rank", "batch_per_gpu": B}}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def attention_fit(samples, esz):
    """Least-squares t = intercept + bytes / rate over the context sweep of the decode attention (bytes = 2 L 1536 esz per launch)."""
    x = np.array([2.0 * s_["context_len"] * 1536 * esz for s_ in samples])
    y = np.array([s_["attn_decode_us"] for s_ in samples])
    if len(x) < 2 or float(np.ptp(x)) == 0.0:
        return None
    slope, icpt = np.polyfit(x, y, 1)
    return {"intercept_us": round(float(icpt), 3), "slope_TBps": round(1.0 / float(slope) / 1e6, 3) if slope > 0 else None,
            "form": "attn_decode_us = intercept_us + bytes / slope"}


def decode_kernel_sweep(dec, L0, T, NS, repeats_mean=6):
    """Per-kind launch times at the run's mean context + the attention kinds as a run average over NS contexts (see main)."""
    mean_L = L0 + (T - 1) / 2.0 + 1.0
    L_ref = int(round(mean_L))
    prof = dec.profile_decode_kernels(repeats=repeats_mean, context_len=L_ref, use_graph=True)
    samples = []
    for i in range(NS):
        L = L0 + 1 + int(round((i + 0.5) * T / NS)) if NS > 1 else L_ref
        p = dec.profile_decode_kernels(repeats=3, context_len=L, use_graph=True)
        samples.append({"context_len": L, "attn_decode_us": round(p["attn_decode"]["avg_us"], 3),
                        "attn_combine_us": round(p["attn_combine"]["avg_us"], 3)})
    at_mean = {k: prof[k]["avg_us"] for k in ("attn_decode", "attn_combine")}
    for k in ("attn_decode", "attn_combine"):      # run average replaces the single point at the mean length
        prof[k]["avg_us"] = float(np.mean([s_[k + "_us"] for s_ in samples]))
    ends = {"at_mean_context": {"context_len": L_ref, "attn_decode_us": round(at_mean["attn_decode"], 3),
                                "attn_combine_us": round(at_mean["attn_combine"], 3)},
            "samples": samples}
    return prof, ends, L_ref, mean_L


def roofline_block(prof, ends, L_ref, mean_L, L0, T, NS, B, precision, decode_only_per_gpu):
    """`roofline` object of a bench line from a kernel sweep: the dominant decode kernel as a run average + the per-kind table."""
    esz = 4 if precision == "fp32" else 2
    per_token_us = {k: v["avg_us"] * LAUNCHES_PER_TOKEN[k] for k, v in prof.items()}
    dom = max(per_token_us, key=per_token_us.get)
    ach = prof[dom]["bytes"] / (prof[dom]["avg_us"] * 1e-6) / 1e9
    bytes_per_token = W_ELEMS * esz / B + KV_ELEMS_PER_POS * mean_L * esz
    names = kernel_names(precision, B > 4)
    traffic = pmc_traffic(dom, names.get(dom, []), L_ref, batched=B > 4, precision=precision)
    layer_us = sum(prof[k]["avg_us"] for k in ("qkv_gemv", "attn_decode", "attn_combine", "out_proj_gemv", "fc1_gemv", "fc2_gemv"))
    return {
        "bound": "hbm", "kernel": dom, "kernel_name": (names.get(dom) or ["?"])[0], "achieved": round(ach, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
        "traffic": traffic.get("bytes"), "traffic_source": traffic.get("source"),
        "bytes_per_launch": prof[dom]["bytes"], "avg_us_per_launch": round(prof[dom]["avg_us"], 3),
        "context_len_at_measurement": L_ref,
        "note": f"run-average: mean launch duration over {NS} contexts spread over the timed run (contexts {L0 + 1}..{L0 + T}) and "
                "the algorithmic bytes of one launch at the mean context length; reproduce with scripts/roofline_from_rocprof.py "
                "on profiles/r06_*_kernel_stats.csv",
        "context_sweep": ends,
        "per_layer_kernel_sum_us": round(layer_us, 2),
        "kernels": {k: {"avg_us": round(v["avg_us"], 3), "GBps": round(v["bytes"] / (v["avg_us"] * 1e-6) / 1e9, 1) if v["avg_us"] > 0 else 0.0,
                        "us_per_token": round(per_token_us[k], 2)} for k, v in prof.items()},
        "whole_step": {"bytes_per_token": bytes_per_token,
                       "achieved_GBps": round(decode_only_per_gpu * bytes_per_token / 1e9, 1),
                       "frac": round(decode_only_per_gpu * bytes_per_token / 1e9 / HBM_PEAK_GBS, 4)},
    }


def first_divergence(ids_a, ids_b):
    """Per row: index of the first position where two id streams differ (len = never)."""
    out = []
    for a, b in zip(ids_a, ids_b):
        a, b = np.asarray(a), np.asarray(b)
        n = min(len(a), len(b))
        d = np.nonzero(a[:n] != b[:n])[0]
        out.append(int(d[0]) if len(d) else n)
    return out


def config3_shard_extra(lmm, dev, points, num_face, T, precision, tok, NS=8):
    """ONE full-size step of BASELINE configs[3]'s per-GPU shard (32 clouds, greedy, T = 4 * num_face tokens) on an existing context,
    under the same clock as the headline: encode + prefill + decode + detokenise (clean=True, the reference's default), with its own
    roofline block.  A 32-token run first builds the B = 32 context (cache reservation, tiled weight copies, the step graph)."""
    import torch
    from edgerunner_amd import weights as W
    Bx = 32
    pcs = torch.cat([W.synthetic_point_cloud(i, points) for i in range(Bx)]).to(dev)
    lmm.generate(pcs, num_face, tokenizer=object(), max_new_tokens=32, min_new_tokens=32)
    torch.cuda.synchronize()
    t = time.perf_counter()
    meshes, toks = lmm.generate(pcs, num_face, tokenizer=tok, max_new_tokens=T, min_new_tokens=T, clean=True)
    torch.cuda.synchronize()
    step_s = time.perf_counter() - t
    dec = lmm.mesh_decoder
    dec_tok_s = Bx * T / (dec.last_decode_ms / 1e3)
    prof, ends, L_ref, mean_L = decode_kernel_sweep(dec, PREFIX, T, NS, repeats_mean=3)
    ends["fit"] = attention_fit(ends["samples"], 4 if precision == "fp32" else 2)
    return {"workload": f"BASELINE configs[3] shard at full size: 32 clouds in one batch on one GPU, greedy, test_num_face={num_face}, {T} new "
                        f"tokens per cloud, {'exact fp32 (ids bit-exact vs the CPU reference)' if precision == 'fp32' else 'fp16 storage / fp32 accumulate'}; "
                        "one step = encode_cond + 2050-token prefill + decode + detokenise (clean=True)",
            "value": round(Bx * T / step_s, 1), "unit": "tokens/s", "ms_per_step": round(step_s * 1e3, 1), "steps": 1,
            "decode_only_tokens_per_s": round(dec_tok_s, 1), "faces_first_mesh": int(len(meshes[0].faces)) if meshes and meshes[0] is not None else None,
            "roofline": roofline_block(prof, ends, L_ref, mean_L, PREFIX, T, NS, Bx, precision, dec_tok_s)}, toks


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if args.dry_run:
        return dry_run(args)

    import torch
    from edgerunner_amd import dist as D
    from edgerunner_amd import weights as W
    from edgerunner_amd.models import LMM
    from edgerunner_amd.options import config_defaults

    rank, world, local = D.init_process_group()
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; reporting the world size actually running", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path (see --dry-run)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    T = args.tokens or 4 * args.num_face
    B = args.batch_per_gpu
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=args.layers, generate_mode=args.mode)

    if world > 1:
        # N ranks generate the same synthetic weights on the host at once (2.7 GB of fp32 each): give every rank its share of the
        # cores instead of N x all of them, so that the set-up of an 8-GPU run is not an oversubscribed host
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    t0 = time.time()
    lmm = LMM(opt, dev, precision=args.precision)
    esz = 4 if args.precision == "fp32" else 2
    keep_sd = rank == 0 and world == 1 and args.cpu_steps > 0 and args.config == 1 and not args.overridden
    sd = {}
    def items():
        for k, t in W.iter_state_dict(opt, 0, "perturbed"):
            if keep_sd:
                sd[k] = t
            yield k, t
    lmm.mesh_decoder.load_state_iter(items(), strict=True)
    if rank == 0:
        print(f"[bench] weights generated + loaded in {time.time() - t0:.1f}s", file=sys.stderr)
    resume = None
    if args.resume_len > 0:      # legal LR_ABSCO prefix: BOM + 9 coordinates, then (L/R + 3 coordinates) groups
        body = [5] + [6 + (i * 37) % 512 for i in range(9)]
        while len(body) < args.resume_len:
            body += [3 + (len(body) // 4) % 2] + [6 + ((len(body) + i) * 53) % 512 for i in range(3)]
        resume = torch.tensor([body[: args.resume_len]] * B, dtype=torch.long)
    n_items = world * B
    # the whole LMM.generate() is timed, detokenise included (core/models.py:309-319, core/provider.py:39-66): ids -> mesh through the
    # native meto engine + the trimesh-style clean-up (clean=True is the reference's default); its cost is also reported by itself
    from edgerunner_amd.meto import Engine, save_mesh
    tok_engine = Engine(opt.discrete_bins, backend=opt.meto_backend)

    def one_step(step_idx):
        mine = D.shard_indices(n_items, rank, world)        # cloud index i runs on rank i mod world
        pcs = torch.cat([W.synthetic_point_cloud(step_idx * n_items + i, args.points) for i in mine]).to(dev)   # resident in HBM
        _, toks = lmm.generate(pcs, args.num_face, tokenizer=tok_engine, max_new_tokens=T, min_new_tokens=T, resume_ids=resume,
                               seed=(1000 + step_idx) if args.mode == "sample" else None, clean=True)
        last_tokens[:] = [toks[0]]
        torch.cuda.synchronize()
        tg = time.perf_counter()
        streams = D.gather_token_streams([t[args.resume_len:] for t in toks], n_items, device=dev)
        gather_ms.append((time.perf_counter() - tg) * 1e3)
        assert len(streams) == n_items and all(len(s) == T for s in streams)
        return lmm.mesh_decoder.last_decode_ms

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.time() - t0:.1f}s] {msg}", file=sys.stderr, flush=True)

    gather_ms = []
    last_tokens = []
    for w in range(args.warmup):
        one_step(-1 - w)
    log("warmup done")
    gather_ms.clear()
    D.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    dec_ms = [one_step(k) for k in range(args.steps)]
    torch.cuda.synchronize()
    D.barrier()
    my_elapsed = time.perf_counter() - t_start
    elapsed = D.max_over_ranks(my_elapsed, dev)
    log(f"timed region done: {elapsed:.2f}s")
    # what a first N > 1 run needs to explain itself: every rank's own wall time, decode rate and gather time (rank order)
    rk_wall = D.all_ranks(my_elapsed, dev)
    rk_dec = D.all_ranks(B * T / (float(np.mean(dec_ms)) / 1e3), dev)
    rk_gather = D.all_ranks(float(np.mean(gather_ms)) if gather_ms else 0.0, dev)
    per_rank = {"wall_s": {"min": round(min(rk_wall), 3), "max": round(max(rk_wall), 3)},
                "decode_tokens_per_s": {"min": round(min(rk_dec), 1), "max": round(max(rk_dec), 1), "by_rank": [round(v, 1) for v in rk_dec]},
                "gather_ms_per_step": {"min": round(min(rk_gather), 3), "max": round(max(rk_gather), 3),
                                       "what": "host wall of the token-stream all-gather (one all-reduce of the width + one all-gather), "
                                               "after the rank's own decode has drained; includes waiting for the slowest rank"}}

    total_tokens = n_items * args.steps * T
    value = total_tokens / elapsed
    decode_only = n_items * T / (D.max_over_ranks(float(np.mean(dec_ms)), dev) / 1e3)

    # ---- roofline of the dominant decode kernel over the run that was timed.  The context grows linearly from
    # L0 + 1 to L0 + T keys over the run.  The weight-streaming kernels do not depend on it; the attention kernels do,
    # and NOT linearly (the balanced kernel's load steps per wave, ceil(L / 16 / 128), step with the context; the version-2
    # kernel's active 128-key workgroups, n = ceil(L / 128) per head, step against the 256 CUs),
    # so their run-average duration - what `rocprofv3 --stats` reports for the same command - is measured as the mean over
    # NS contexts spread evenly over the run (midpoints of NS equal segments), each a hipGraph replay of the 24 launches of
    # a kind with HIP events on the launch stream (all 24 layers' data, so nothing is cache-resident).  `achieved` =
    # algorithmic bytes of one launch at the MEAN context length / that mean duration.
    # scripts/roofline_from_rocprof.py recomputes `frac` from profiles/*_kernel_stats.csv and checks the two agree.
    L0 = PREFIX + args.resume_len
    NS = 16 if T >= 64 else 1
    prof, ends, L_ref, mean_L = decode_kernel_sweep(lmm.mesh_decoder, L0, T, NS)
    ends["fit"] = attention_fit(ends["samples"], esz)
    log("kernel sweep done")
    roofline = roofline_block(prof, ends, L_ref, mean_L, L0, T, NS, B, args.precision, decode_only / world)
    # detokenise by itself (host): the stream of the last timed step, both forms of the clean flag
    detok = {}
    if last_tokens:
        for clean in (False, True):
            td = time.perf_counter()
            for _ in range(5):
                save_mesh(last_tokens[0], opt, tokenizer=tok_engine, clean=clean)
            detok["clean" if clean else "raw"] = round((time.perf_counter() - td) / 5 * 1e3, 3)

    cfg_name = CONFIGS[args.config]["name"] + (f" with {', '.join(args.overridden)} overridden" if args.overridden else "")
    out = {
        "metric": f"mesh tokens/sec (whole node), ArAE {args.mode} test_num_face={args.num_face}",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "world_size_seen": world, "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f16 storage / f32 accumulate", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: ArAE random-init (seeded synthetic checkpoint), batch {B} per GPU, "
                               f"{'greedy' if args.mode == 'greedy' else 'sample mode (top-k 10, Philox inverse-CDF draw on the device)'}, "
                               f"test_num_face={args.num_face}, {T} new tokens (EOS suppressed until T), "
                               f"{args.points}-point synthetic cloud; step = the whole LMM.generate(): encode_cond + {L0}-token prefill + "
                               f"{T}-token decode + detokenise to a mesh (native meto engine, clean=True)"
                               f"{' + RCCL all-gather of token streams' if world > 1 else ''}",
                   "baseline_config": args.config, "layers": args.layers, "hidden": 1536, "heads": 16, "tokens_per_sample": T,
                   "batch_per_gpu": B, "generate_mode": args.mode, "precision": args.precision,
                   "parallelism": f"dp{world} (independent samples, full replica per GPU)"},
        "decode_only_tokens_per_s": round(decode_only, 2),
        "detokenise_ms": {"per_sample_clean_true": detok.get("clean"), "per_sample_clean_false": detok.get("raw"),
                          "note": "host time of save_mesh (ids -> vertices / faces, core/provider.py:39-66) for one 4000-token stream; INSIDE the timed "
                                  "step with clean=True"},
        "per_rank": per_rank,
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and args.config == 1 and not args.overridden and not args.no_fast_extra:
        # BASELINE configs[3]'s shard (32 clouds per GPU) at FULL size in the exact mode - the mode whose greedy ids are bit-exact vs the
        # CPU reference (tests/test_gpu_parity.py::test_config4_shard_B32_T4000_row0_bit_exact) - on the headline's own context
        exact_toks = None
        try:
            out["config3_shard_exact_fp32"], exact_toks = config3_shard_extra(lmm, dev, args.points, args.num_face, T, "fp32", tok_engine)
        except Exception as e:  # noqa: BLE001 - a secondary figure must never cost the bench line
            out["config3_shard_exact_fp32"] = {"error": repr(e)[:200]}
        log("configs[3] shard, exact mode, done")
        # secondary figure (not `value`): the same workload in the fp16-storage fast mode, the reference's GPU dtype
        del lmm
        torch.cuda.empty_cache()
        fast = LMM(opt, dev, precision="fp16")
        fast.mesh_decoder.load_state_iter(W.iter_state_dict(opt, 0, "perturbed"), strict=True)
        pc = W.synthetic_point_cloud(0, args.points).to(dev)
        for _ in range(2):
            fast.generate(pc, args.num_face, tokenizer=object(), max_new_tokens=T, min_new_tokens=T)
        fb = W_ELEMS * 2 + KV_ELEMS_PER_POS * mean_L * 2
        ftok = T / (fast.mesh_decoder.last_decode_ms / 1e3)
        out["fast_mode_fp16"] = {"decode_only_tokens_per_s": round(ftok, 2), "bytes_per_token": fb,
                                 "hbm_frac": round(ftok * fb / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "fp16 weights + fp16 KV, fp32 accumulate; parity = ids exact / logits 5e-6 vs the "
                                         "oracle on fp16-rounded storage (tests/test_gpu_parity.py), ~1e-3 vs fp32"}
        try:        # the fast mode's own per-kind table and context sweep: same method as `roofline` above
            fprof, fends, fL, _ = decode_kernel_sweep(fast.mesh_decoder, L0, T, NS)
            fends["fit"] = attention_fit(fends["samples"], 2)
            fnames = kernel_names("fp16", False)
            ftr = pmc_traffic("attn_decode", fnames.get("attn_decode", []), fL, batched=False, precision="fp16")
            out["fast_mode_fp16"].update({
                "context_len_at_measurement": fL, "context_sweep": fends,
                "attention": {"kernel_name": (fnames.get("attn_decode") or ["?"])[0], "bytes_per_launch": fprof["attn_decode"]["bytes"],
                              "avg_us_per_launch": round(fprof["attn_decode"]["avg_us"], 3),
                              "frac": round(fprof["attn_decode"]["bytes"] / (fprof["attn_decode"]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                              "traffic": ftr.get("bytes"), "traffic_source": ftr.get("source")},
                "per_layer_kernel_sum_us": round(sum(fprof[k]["avg_us"] for k in ("qkv_gemv", "attn_decode", "attn_combine", "out_proj_gemv",
                                                                                  "fc1_gemv", "fc2_gemv")), 2),
                "kernels": {k: {"avg_us": round(v["avg_us"], 3), "kernel_name": (fnames.get(k) or ["?"])[0],
                                "GBps": round(v["bytes"] / (v["avg_us"] * 1e-6) / 1e9, 1) if v["avg_us"] > 0 else 0.0,
                                "us_per_token": round(v["avg_us"] * LAUNCHES_PER_TOKEN[k], 2)} for k, v in fprof.items()}})
        except Exception as e:  # noqa: BLE001 - a secondary figure must never cost the bench line
            out["fast_mode_fp16"]["kernels_error"] = repr(e)[:200]
        # the batched shard of BASELINE configs[3] (32 independent clouds per GPU) at FULL size in the fast mode, and how far its greedy
        # streams follow the exact mode's on the same clouds (how bit-exact the mode is that configs[3] is quoted in)
        try:
            out["config3_shard_fp16"], fast_toks = config3_shard_extra(fast, dev, args.points, args.num_face, T, "fp16", tok_engine)
            if exact_toks is not None:
                fd = first_divergence(fast_toks, exact_toks)
                out["config3_shard_fp16"]["greedy_ids_vs_exact_fp32"] = {
                    "first_divergence_index_per_row": {"min": int(min(fd)), "median": int(np.median(fd)), "max": int(max(fd))},
                    "rows_identical_over_T": int(sum(1 for v in fd if v >= T)), "rows": len(fd),
                    "note": "index of the first greedy id that differs from the exact-fp32 run of the same cloud (T = never): fp16 storage "
                            "moves logits by ~1e-3, a near-tie flips an arg-max and the streams separate from there"}
        except Exception as e:  # noqa: BLE001 - a secondary figure must never cost the bench line
            out["config3_shard_fp16"] = {"error": repr(e)[:200]}
        log("configs[3] shard, fast mode, done")
        # BASELINE configs[2]'s shape (B = 32, SAMPLE mode top-k 10, test_num_face = 4000, fp16) at a reduced length: the
        # full T = 16000 run takes 143 s (profiles/r02_config3_B32_T16000_fp16_sample.log: 3.58k tok/s = 68 % of 8 TB/s)
        try:
            Bx, Tx = 32, 1024
            sopt = dataclasses.replace(opt, generate_mode="sample")
            fast.opt = sopt
            pcs = torch.cat([W.synthetic_point_cloud(i, args.points) for i in range(Bx)]).to(dev)
            ids = fast.generate_ids(pcs, 4000, tokenizer=object(), max_new_tokens=Tx, min_new_tokens=Tx, seed=1)
            bms = fast.mesh_decoder.last_decode_ms
            bb = W_ELEMS * 2 + Bx * KV_ELEMS_PER_POS * (2050 + (Tx - 1) / 2.0 + 1) * 2
            out["config2_shape_sample_fp16"] = {
                "aggregate_decode_tokens_per_s": round(Bx * Tx / bms * 1e3, 1), "tokens_per_row": Tx, "mode": "sample top_k=10",
                "distinct_rows": len({tuple(r) for r in ids.cpu().numpy()[:, :64].tolist()}),
                "hbm_frac": round(bb / (bms / Tx * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "BASELINE configs[2] shape at reduced length (context 2050..3074); full size T = 16000: "
                        "profiles/r02_config3_B32_T16000_fp16_sample.log"}
            fast.opt = opt
        except Exception as e:  # noqa: BLE001
            out["config2_shape_sample_fp16"] = {"error": repr(e)[:200]}
        del fast
        log("fast-mode + configs[2]-shape passes done")
        try:
            torch.cuda.empty_cache()
            out["dit_front_end_fp16"] = dit_front_end_extra(dev)
        except Exception as e:  # noqa: BLE001 - a secondary figure must never cost the bench line
            out["dit_front_end_fp16"] = {"error": repr(e)[:200]}
        log("DiT front-end pass done")
    if keep_sd:
        out["cpu_baseline"] = cpu_baseline(opt, sd, args.cpu_steps, args.points)
        cb = out["cpu_baseline"]
        out["gpu_over_cpu"] = {
            "vs_sample_on_this_box": round(out["decode_only_tokens_per_s"] / cb["value"], 1),
            "vs_full_run": round(out["value"] / cb["full_run"]["end_to_end_tokens_per_s"], 1) if cb.get("full_run") else None,
            "note": "vs_sample_on_this_box = GPU decode-only rate over the full run / CPU decode rate of the bounded sample at contexts "
                    "2050..2170 (THIS box's host cores: conservative, the CPU slows down as the context grows); vs_full_run = `value` "
                    "(encode + prefill + 4000-token decode) / the committed full three-phase CPU run (build container, 8 threads). "
                    "A GPU/CPU ratio says nothing about kernel quality - the roofline fraction does."}
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
