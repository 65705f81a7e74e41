"""Row a14: the drop-in scripts themselves on the GPU - ``infer.py`` (reference infer.py:34-137) and
``infer_dit.py`` (reference infer_dit.py:40-135) are run as subprocesses on a tiny synthetic checkpoint written to disk
(ArAE widths, 2 decoder layers) and their ``*_tokens.npy`` outputs (ids - 3, cut at EOS: infer.py:113-116) are compared
with the CPU oracle.  ``model.half()`` inside the scripts must really select the fp16-storage context."""
import dataclasses
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def run_script(script, args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("ER_NO_GRAPH", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + [str(a) for a in args], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:]
    return p.stdout


def trimmed(ids):
    from edgerunner_amd.utils import trim_tokens
    return trim_tokens(np.asarray(ids))


@pytest.fixture(scope="module")
def arae_ckpt(tmp_path_factory):
    from safetensors.torch import save_file
    from edgerunner_amd import weights as W
    from edgerunner_amd.options import config_defaults
    d = tmp_path_factory.mktemp("ckpt")
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=2, generate_mode="greedy")
    sd = W.make_state_dict(opt, 0, "perturbed")
    path = str(d / "arae_2layers.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    return opt, sd, path


def test_infer_py_batched_jobs_half_and_float(arae_ckpt, tmp_path):
    import arae_oracle as O
    from edgerunner_amd import weights as W
    opt, sd, ckpt = arae_ckpt
    inp = tmp_path / "inputs"
    inp.mkdir()
    clouds = {"a": W.synthetic_point_cloud(0, 512), "b": W.synthetic_point_cloud(3, 512), "c": W.synthetic_point_cloud(4, 300)}
    for n, pc in clouds.items():
        np.save(inp / f"{n}.npy", pc[0].numpy())
    T = 40
    # ---- default precision: the script's model.half() must select the fp16-storage context
    out = tmp_path / "out_half"
    log = run_script("infer.py", ["ArAE", "--num_layers", 2, "--resume", ckpt, "--test_path", inp, "--generate_mode", "greedy",
                                  "--test_num_face", 1000, 4000, "--test_repeat", 2, "--test_max_seq_length", T,
                                  "--workspace", out])
    assert "jobs in this call" in log
    assert any(f"({n} jobs in this call)" in log for n in (4, 3, 2)), log[-2000:]      # jobs were batched, not run one by one
    sd16 = O.round_streamed_weights(sd, torch.float16)
    fwd16 = O.make_forward(sd16, opt, kv_round=torch.float16)
    allz = dict(np.load(out / "tokens_all.npz"))
    assert len(allz) == 3 * 2 * 2
    for n, pc in clouds.items():
        assert (out / f"{n}_pc.obj").exists()
        for nf in (1000, 4000):
            want = trimmed(O.lmm_generate_ids(sd16, opt, pc, nf, max_new_tokens=T, fwd=fwd16).numpy()[0])
            for rep in (0, 1):
                got = np.load(out / f"{n}_{rep}_{nf}f_tokens.npy")
                assert np.array_equal(got, want), (n, nf, rep, got, want)
                assert np.array_equal(allz[f"{n}_{rep}_{nf}f"], want)
                assert (out / f"{n}_{rep}_{nf}f.ply").exists()
    # ---- exact mode requested from the environment: .float() path, ids == the fp32 CPU run
    out32 = tmp_path / "out_f32"
    run_script("infer.py", ["ArAE", "--num_layers", 2, "--resume", ckpt, "--test_path", inp / "a.npy", "--generate_mode", "greedy",
                            "--test_num_face", 1000, "--test_max_seq_length", T, "--workspace", out32],
               {"EDGERUNNER_PRECISION": "fp32"})
    want = trimmed(O.lmm_generate_ids(sd, opt, clouds["a"], 1000, max_new_tokens=T).numpy()[0])
    assert np.array_equal(np.load(out32 / "a_0_1000f_tokens.npy"), want)
    # ---- sample mode: repeats of one input must differ (per-row Philox streams), and a second run reproduces them (--seed)
    outs = tmp_path / "out_sample"
    for k in (0, 1):
        run_script("infer.py", ["ArAE", "--num_layers", 2, "--resume", ckpt, "--test_path", inp / "a.npy", "--generate_mode", "sample",
                                "--test_num_face", 1000, "--test_repeat", 3, "--test_max_seq_length", T, "--seed", 5,
                                "--workspace", f"{outs}{k}"])
    r = [[np.load(f"{outs}{k}/a_{i}_1000f_tokens.npy") for i in range(3)] for k in (0, 1)]
    assert all(np.array_equal(r[0][i], r[1][i]) for i in range(3)), "same --seed must reproduce the samples"
    assert len({tuple(x.tolist()) for x in r[0]}) > 1, "test_repeat samples must differ"


def test_infer_py_sample_mode_does_not_depend_on_job_grouping(arae_ckpt, tmp_path):
    """a job's sampled tokens are a function of (--seed, job index) only - the Philox stream id of a row is the
    job's index in the reference's loop order (infer.py:99-101), not its position in whatever batch it landed in.  Exact
    mode, so the batched and the single-row kernels agree bit for bit: one job per call vs all six jobs in one call."""
    from edgerunner_amd import weights as W
    opt, sd, ckpt = arae_ckpt
    inp = tmp_path / "a.npy"
    np.save(inp, W.synthetic_point_cloud(0, 512)[0].numpy())
    runs = {}
    for name, nb in (("one", "1"), ("three", "3"), ("all", "32")):
        out = tmp_path / f"out_{name}"
        log = run_script("infer.py", ["ArAE", "--num_layers", 2, "--resume", ckpt, "--test_path", inp, "--generate_mode", "sample",
                                      "--test_num_face", 1000, 4000, "--test_repeat", 3, "--test_max_seq_length", 48, "--seed", 9,
                                      "--workspace", out], {"EDGERUNNER_PRECISION": "fp32", "ER_INFER_BATCH": nb})
        assert ("(1 jobs in this call)" in log) == (nb == "1"), log[-1500:]
        runs[name] = {k: v for k, v in np.load(out / "tokens_all.npz").items()}
    assert len(runs["one"]) == 6
    for k, v in runs["one"].items():
        assert np.array_equal(v, runs["three"][k]) and np.array_equal(v, runs["all"][k]), k
    assert len({tuple(v.tolist()) for v in runs["one"].values()}) > 3, "six jobs, six Philox streams"


def test_infer_py_cond_mode_none(tmp_path):
    """reference infer.py:96-97 / core/models.py:131-141: cond_mode='none' generates from the face-count token alone
    (num_cond_tokens = 1), once per input path; ids == the CPU oracle."""
    import arae_oracle as O
    from safetensors.torch import save_file
    from edgerunner_amd import weights as W
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=2, generate_mode="greedy", cond_mode="none", num_cond_tokens=1)
    sd = W.make_state_dict(opt, 0, "perturbed")
    assert not any(k.startswith(("point_encoder.", "proj_cond.")) for k in sd)
    ckpt = str(tmp_path / "none.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, ckpt)
    (tmp_path / "inputs").mkdir()
    for n in ("x", "y"):
        np.save(tmp_path / "inputs" / f"{n}.npy", np.zeros((4, 3), np.float32))      # the path only names the outputs
    out = tmp_path / "out"
    run_script("infer.py", ["ArAE", "--num_layers", 2, "--cond_mode", "none", "--num_cond_tokens", 1, "--resume", ckpt,
                            "--test_path", tmp_path / "inputs", "--generate_mode", "greedy", "--test_num_face", 1000, 8000,
                            "--test_max_seq_length", 40, "--workspace", out], {"EDGERUNNER_PRECISION": "fp32"})
    for nf in (1000, 8000):
        want = trimmed(O.lmm_generate_ids(sd, opt, torch.zeros(1, 0), nf, max_new_tokens=40).numpy()[0])
        for n in ("x", "y"):
            assert np.array_equal(np.load(out / f"{n}_0_{nf}f_tokens.npy"), want), (n, nf)
        assert not (out / "x_pc.obj").exists()


def test_rccl_code_path_on_one_gpu(arae_ckpt, tmp_path):
    """The N > 1 exchange (dist.py: all-reduce of the width, all_gather_into_tensor of the packed streams, barrier, max-over-ranks)
    has no multi-GPU box to run on from here; ER_DIST_FORCE=1 keeps those collectives ON in a ONE-rank `nccl` (= RCCL) group, so at
    least the RCCL initialisation, dtypes and device placement of every call execute on real hardware: infer.py end to end, then the
    gathered archive must equal the per-job files."""
    from edgerunner_amd import weights as W
    opt, sd, ckpt = arae_ckpt
    inp = tmp_path / "a.npy"
    np.save(inp, W.synthetic_point_cloud(0, 512)[0].numpy())
    out = tmp_path / "out"
    env = {"ER_DIST_FORCE": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533",
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    log = run_script("infer.py", ["ArAE", "--num_layers", 2, "--resume", ckpt, "--test_path", inp, "--generate_mode", "greedy",
                                  "--test_num_face", 1000, 2000, "--test_max_seq_length", 32, "--workspace", out], env)
    assert "token streams gathered" in log
    allz = dict(np.load(out / "tokens_all.npz"))
    for nf in (1000, 2000):
        assert np.array_equal(allz[f"a_0_{nf}f"], np.load(out / f"a_0_{nf}f_tokens.npy"))
    # bench.py's plumbing (barrier + max-over-ranks + gather) in the same one-rank RCCL group
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--layers", "2", "--tokens", "32",
                        "--cpu-steps", "0", "--no-fast-extra"], env={**os.environ, **env, "MASTER_PORT": "29534"}, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    assert '"n_gpus": 1' in line and '"value"' in line


def test_lmm_half_selects_fp16_context_or_fails_loudly(arae_ckpt):
    from edgerunner_amd import native
    from edgerunner_amd.models import LMM
    opt, sd, _ = arae_ckpt
    m = LMM(opt, DEV, precision=None)
    m.load_state_dict(sd, strict=True)
    assert m.precision == "fp32"
    m = m.half().eval().to(DEV)
    assert m.precision == "fp16" and m.mesh_decoder is not None
    pc = __import__("edgerunner_amd.weights", fromlist=["x"]).synthetic_point_cloud(0, 256).to(DEV)
    _, t16 = m.generate(pc, 1000, tokenizer=object(), max_new_tokens=16, min_new_tokens=16)
    m = m.float()                                   # rebuilds the context from the retained checkpoint
    assert m.precision == "fp32"
    _, t32 = m.generate(pc, 1000, tokenizer=object(), max_new_tokens=16, min_new_tokens=16)
    assert len(t16[0]) == len(t32[0]) == 16
    eager = LMM(opt, DEV, precision="fp32")
    eager.mesh_decoder.load_state_iter(sd.items(), strict=True)     # streamed straight into the context: not replayable
    with pytest.raises(native.NativeError, match="half"):
        eager.half()


def test_sample_mode_default_seed_advances_between_calls(arae_ckpt):
    """without an explicit seed, two generate() calls must not replay the same Philox stream (the reference
    draws from torch's global generator, which advances), while torch.manual_seed() reproduces the pair."""
    from edgerunner_amd import weights as W
    from edgerunner_amd.models import LMM
    opt, sd, _ = arae_ckpt
    sopt = dataclasses.replace(opt, generate_mode="sample")
    m = LMM(sopt, DEV)
    m.load_state_dict(sd, strict=True)
    pc = W.synthetic_point_cloud(0, 256).to(DEV)

    def pair():
        torch.manual_seed(123)
        a = m.generate_ids(pc, 1000, tokenizer=object(), max_new_tokens=24, min_new_tokens=24).cpu()
        b = m.generate_ids(pc, 1000, tokenizer=object(), max_new_tokens=24, min_new_tokens=24).cpu()
        return a, b
    a1, b1 = pair()
    a2, b2 = pair()
    assert not torch.equal(a1, b1)
    assert torch.equal(a1, a2) and torch.equal(b1, b2)


def test_infer_dit_py_image_to_tokens(tmp_path):
    """infer_dit.py at reduced depth (2 DiT / 2 CLIP / 2 decoder layers, 3 DDIM steps) in the exact mode: the
    ``*_tokens.npy`` it writes must equal image -> CLIP -> DiT sampler -> ArAE greedy decode of the CPU oracle, given the
    same initial noise (the script draws it on the device right after seed_everything(seed): replayed here)."""
    import arae_oracle as O
    from safetensors.torch import save_file
    from edgerunner_amd import weights as W
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["DiT"], num_layers=2, dit_num_layers=2, generate_mode="greedy")
    lat_opt = dataclasses.replace(opt, cond_mode="point_latent")
    sd_l = W.make_state_dict(lat_opt, 0, "perturbed")
    sd_d = W.make_dit_state_dict(opt, 0, "perturbed")
    sd_d.update(W.make_clip_state_dict(2, 0, "perturbed"))
    ck1, ck2 = str(tmp_path / "lmm.safetensors"), str(tmp_path / "dit.safetensors")
    save_file({k: v.contiguous() for k, v in sd_l.items()}, ck1)
    save_file({k: v.contiguous() for k, v in sd_d.items()}, ck2)
    g = torch.Generator().manual_seed(7)
    img = torch.rand(96, 80, 3, generator=g)
    np.save(tmp_path / "img.npy", img.numpy())
    T, seed = 24, 11
    out = tmp_path / "out"
    env = {"EDGERUNNER_PRECISION": "fp32", "ER_CLIP_LAYERS": "2", "ER_DIT_STEPS": "3"}
    run_script("infer_dit.py", ["DiT", "--num_layers", 2, "--dit_num_layers", 2, "--resume", ck1, "--resume2", ck2, "--test_path",
                                tmp_path / "img.npy", "--generate_mode", "greedy", "--test_num_face", 1000,
                                "--test_max_seq_length", T, "--seed", seed, "--workspace", out], env)
    got = np.load(out / "img_0_1000f_tokens.npy")
    torch.cuda.manual_seed_all(seed)
    noise = torch.randn(1, 2048, 64, device=DEV, dtype=torch.float32).cpu()
    cond_img = torch.nn.functional.interpolate(img.permute(2, 0, 1)[None], (512, 512), mode="bilinear", align_corners=False)
    cond = O.mdit_get_cond(sd_d, cond_img)
    lat = O.mdit_run(sd_d, cond, noise, opt.dit_num_heads, num_inference_steps=3, guidance_scale=7.5)
    want = trimmed(O.lmm_generate_ids(sd_l, lat_opt, lat, 1000, max_new_tokens=T).numpy()[0])
    assert np.array_equal(got, want), (got, want)
    # default precision (.half() on both models): runs and writes a grammatical stream
    out16 = tmp_path / "out16"
    run_script("infer_dit.py", ["DiT", "--num_layers", 2, "--dit_num_layers", 2, "--resume", ck1, "--resume2", ck2, "--test_path",
                                tmp_path / "img.npy", "--generate_mode", "greedy", "--test_num_face", 1000,
                                "--test_max_seq_length", T, "--seed", seed, "--workspace", out16],
               {"ER_CLIP_LAYERS": "2", "ER_DIT_STEPS": "3"})
    t16 = np.load(out16 / "img_0_1000f_tokens.npy")
    assert t16.ndim == 1 and (len(t16) == 0 or t16[0] == 2)      # first token is BOM (5 - 3)
