"""DiT image-conditioned front-end (scope row f3) through the C ABI (er_dit_*) against goldens produced by the
reference's own DiT module (oracle/make_golden.py dit) and against the oracle live.  fp32: 1e-3 on latents
(they are O(1); a 6-step CFG sampler amplifies round-off), ids of the downstream ArAE decode exact."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_small.npz")


@pytest.fixture(scope="module")
def setup():
    from edgerunner_amd import weights as W
    from edgerunner_amd.models_dit import MDiT
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=2, generate_mode="greedy", cond_mode="point_latent",
                              dit_num_layers=2)
    sd = W.make_dit_state_dict(opt, 0, "perturbed")
    m = MDiT(opt, DEV, clip_layers=0)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    g = dict(np.load(GOLD))
    gen = torch.Generator().manual_seed(int(g["seed"][0]))
    clip_hidden = torch.randn(1, 257, 1280, generator=gen)
    noise = torch.randn(1, 2048, 64, generator=gen)
    x = torch.randn(2, 2048, 64, generator=gen)
    return opt, sd, m, g, clip_hidden, noise, x


def test_project_cond_and_forward_vs_reference_module(setup):
    opt, sd, m, g, clip_hidden, noise, x = setup
    cond = m.get_cond(clip_hidden.to(DEV))
    err = np.abs(cond[0, [0, 100, 256]].cpu().numpy() - g["cond_rows"]).max()
    assert err < 1e-4, err
    c2 = torch.cat([torch.zeros_like(cond), cond])
    y = m.dit(x.to(DEV), c2, torch.tensor(g["t"]))
    rows = g["rows"].tolist()
    err = np.abs(y[:, rows].cpu().numpy() - g["fwd_rows"]).max()
    rel = abs(float(y.double().sum()) - g["fwd_sum"][0]) / g["fwd_sum"][1]
    print(f"DiT forward: max abs err on sampled rows {err:.3e}, checksum rel err {rel:.3e}")
    assert err < 1e-3 and rel < 1e-5


def test_sampler_vs_reference_and_downstream_decode(setup):
    opt, sd, m, g, clip_hidden, noise, x = setup
    lat = m.run(clip_hidden.to(DEV), num_inference_steps=6, guidance_scale=7.5, noise=noise.to(DEV))
    rows = g["rows"].tolist()
    err = np.abs(lat[0, rows].cpu().numpy() - g["lat_rows"]).max()
    rel = abs(float(lat.double().sum()) - g["lat_sum"][0]) / g["lat_sum"][1]
    print(f"6-step CFG/DDIM latents: max abs err {err:.3e}, checksum rel err {rel:.3e}")
    assert err < 2e-3 and rel < 1e-4
    # latents -> ArAE decode in point_latent mode (infer_dit.py:111-113)
    from edgerunner_amd import weights as W
    from edgerunner_amd.models import LMM
    lmm = LMM(opt, DEV)
    lmm.mesh_decoder.load_state_iter(W.iter_state_dict(opt, 0, "perturbed"), strict=True)
    _, toks = lmm.generate(lat, 1000, tokenizer=object(), max_new_tokens=32, min_new_tokens=32)
    assert np.array_equal(toks[0], g["ids_from_latents"][0]), (toks[0], g["ids_from_latents"][0])


def test_sampler_vs_oracle_live_batch2(setup):
    import arae_oracle as O
    opt, sd, m, g, clip_hidden, noise, x = setup
    gen = torch.Generator().manual_seed(99)
    ch = torch.randn(2, 257, 1280, generator=gen)
    nz = torch.randn(2, 2048, 64, generator=gen)
    want = O.mdit_run(sd, O.dit_project_cond(sd, ch), nz, opt.dit_num_heads, num_inference_steps=3, guidance_scale=5.0)
    got = m.run(ch.to(DEV), num_inference_steps=3, guidance_scale=5.0, noise=nz.to(DEV)).cpu()
    err = float((got - want).abs().max())
    print(f"3-step sampler, batch 2: max abs err {err:.3e}")
    assert err < 2e-3


def test_clip_image_encoder_and_image_to_tokens(setup):
    """BASELINE configs[4] end to end at reduced depth: image -> CLIP ViT (ViT-H/14 widths, 2 layers) -> proj/norm ->
    DiT (2 layers) sampled with DDIM + CFG -> latents -> ArAE greedy decode (2 layers), vs the CPU oracle."""
    import arae_oracle as O
    from edgerunner_amd import weights as W
    from edgerunner_amd.models import LMM
    from edgerunner_amd.models_dit import MDiT
    opt, sd, _, g, clip_hidden, noise, x = setup
    sd_all = dict(sd)
    sd_all.update(W.make_clip_state_dict(2, 0, "perturbed"))
    m = MDiT(opt, DEV, clip_layers=2)
    m.load_state_dict(sd_all, strict=True)
    gen = torch.Generator().manual_seed(int(g["seed"][0]))
    for shape in ((1, 257, 1280), (1, 2048, 64), (2, 2048, 64)):     # replay the generator up to the golden image
        torch.randn(*shape, generator=gen)
    img = torch.rand(1, 3, 512, 512, generator=gen)
    # (a) hidden states vs the installed-transformers golden
    m_hid = torch.empty(0)
    cond = m.get_cond(img.to(DEV))
    want_cond = O.mdit_get_cond(sd_all, img)
    err = float((cond.cpu() - want_cond).abs().max())
    print(f"image -> cond: max abs err vs oracle {err:.3e}")
    assert err < 2e-4
    hid_rows = O.clip_vision_forward(sd_all, O.clip_preprocess(img))[0, [0, 1, 128, 256]].numpy()
    assert np.abs(hid_rows - g["clip_rows"]).max() < 1e-4          # oracle restatement vs installed transformers golden
    # (b) odd image size (bilinear resize path) vs oracle
    img2 = torch.rand(2, 3, 300, 417, generator=gen)
    err2 = float((m.get_cond(img2.to(DEV)).cpu() - O.mdit_get_cond(sd_all, img2)).abs().max())
    assert err2 < 2e-4, err2
    # (c) image -> latents -> tokens
    nz = torch.randn(1, 2048, 64, generator=gen)
    lat = m.run(img.to(DEV), num_inference_steps=4, guidance_scale=7.5, noise=nz.to(DEV))
    want_lat = O.mdit_run(sd_all, want_cond, nz, opt.dit_num_heads, num_inference_steps=4, guidance_scale=7.5)
    errl = float((lat.cpu() - want_lat).abs().max())
    print(f"image -> latents (4 steps): max abs err {errl:.3e}")
    assert errl < 2e-3
    lmm = LMM(opt, DEV)
    sd_l = W.make_state_dict(opt, 0, "perturbed")
    lmm.load_state_dict(sd_l, strict=True)
    _, toks = lmm.generate(lat, 1000, tokenizer=object(), max_new_tokens=24, min_new_tokens=24)
    want_ids = O.lmm_generate_ids(sd_l, opt, want_lat, 1000, max_new_tokens=24, min_new_tokens=24).numpy()[0]
    assert np.array_equal(toks[0], want_ids), (toks[0], want_ids)


def test_unbuilt_pieces_fail_loudly(setup):
    opt, sd, m, g, clip_hidden, noise, x = setup
    with pytest.raises(NotImplementedError, match="image encoder"):
        m.get_cond(torch.rand(1, 3, 512, 512))
    with pytest.raises(IndexError):                     # reference: scheduler.timesteps[int(steps * strength)] out of range
        m.run(clip_hidden.to(DEV), num_inference_steps=4, latents=noise.to(DEV), strength=1.0)
    with pytest.raises(ValueError, match="latents must be"):
        m.run(clip_hidden.to(DEV), num_inference_steps=4, latents=noise[:, :100].to(DEV))


def test_img2img_branch_and_num_repeat(setup):
    """MDiT.run with latents given (core/models_dit.py:207-209: add_noise at timesteps[int(steps * strength)], loop
    over the remaining timesteps) and num_repeat = 2 (cond repeat_interleave, :203), vs the oracle."""
    import arae_oracle as O
    opt, sd, m, g, clip_hidden, noise, x = setup
    gen = torch.Generator().manual_seed(123)
    ch = torch.randn(1, 257, 1280, generator=gen)
    nz = torch.randn(2, 2048, 64, generator=gen)
    lat0 = 0.5 * torch.randn(2, 2048, 64, generator=gen)
    cond2 = O.dit_project_cond(sd, ch).repeat_interleave(2, dim=0)
    for strength, steps in ((0.5, 6), (0.0, 3)):
        want = O.mdit_run(sd, cond2, nz, opt.dit_num_heads, num_inference_steps=steps, guidance_scale=3.0, latents=lat0,
                          strength=strength)
        got = m.run(ch.to(DEV), num_inference_steps=steps, guidance_scale=3.0, num_repeat=2, latents=lat0.to(DEV),
                    strength=strength, noise=nz.to(DEV)).cpu()
        err = float((got - want).abs().max())
        print(f"img2img strength {strength}, {steps} steps, num_repeat 2: max abs err {err:.3e}")
        assert err < 2e-3


def test_fast_mode_fp16_mfma_vs_emulation(setup):
    """precision='fp16': every Linear of CLIP / proj_cond / DiT on the fp16-input matrix cores.  Must match the oracle
    with fp16-rounded weights AND fp16-rounded Linear inputs (fp32 math otherwise); distance to fp32 is reported."""
    import arae_oracle as O
    from edgerunner_amd import weights as W
    from edgerunner_amd.models_dit import MDiT
    opt, sd, m32, g, clip_hidden, noise, x = setup
    sd_all = dict(sd)
    sd_all.update(W.make_clip_state_dict(2, 0, "perturbed"))
    m = MDiT(opt, DEV, clip_layers=2, precision="fp16")
    m.load_state_dict(sd_all, strict=True)
    sd_h = O.round_linear_weights(sd_all)
    gen = torch.Generator().manual_seed(2024)
    img = torch.rand(1, 3, 384, 384, generator=gen)
    nz = torch.randn(1, 2048, 64, generator=gen)
    with O.linear_input_rounding(torch.float16):
        want_cond = O.mdit_get_cond(sd_h, img)
        want_lat = O.mdit_run(sd_h, want_cond, nz, opt.dit_num_heads, num_inference_steps=4, guidance_scale=7.5)
    ref_lat = O.mdit_run(sd_all, O.mdit_get_cond(sd_all, img), nz, opt.dit_num_heads, num_inference_steps=4, guidance_scale=7.5)
    cond = m.get_cond(img.to(DEV))
    lat = m.run(img.to(DEV), num_inference_steps=4, guidance_scale=7.5, noise=nz.to(DEV))
    e_cond = float((cond.cpu() - want_cond).abs().max())
    e_lat = float((lat.cpu() - want_lat).abs().max())
    drift = float((lat.cpu() - ref_lat).abs().max())
    print(f"fp16 MFMA front-end: cond err vs emulation {e_cond:.3e}, latents err {e_lat:.3e}; latents vs fp32 path {drift:.3e}")
    # the kernel's fp16 roundings (Linear inputs, q/k/v, p) sit on rounding boundaries the emulation's slightly
    # different fp32 sums can flip; 4 CFG-7.5 steps amplify those flips to a few 1e-3 on O(1) latents
    assert e_cond < 2e-3 and e_lat < 1e-2 and drift < 5e-2


def test_fast_mode_fp16_latent_size_not_a_multiple_of_64():
    """fp16 mode with a latent size that is not a whole number of 64-token tiles (point_latent_size = 100): the LDS-DMA
    GEMM / attention pair does not apply, the context must fall back to the fp32-operand fused attention + register-staged
    fp16 GEMMs (round-3 advisor: that shape returned ER_ERR_INVALID) and still match the fp16 emulation."""
    import arae_oracle as O
    from edgerunner_amd import weights as W
    from edgerunner_amd.models_dit import MDiT
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=2, generate_mode="greedy", cond_mode="point_latent",
                              dit_num_layers=2, point_latent_size=100)
    sd = W.make_dit_state_dict(opt, 3, "perturbed")
    m = MDiT(opt, DEV, clip_layers=0, precision="fp16")
    m.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(77)
    clip_hidden = torch.randn(2, 257, 1280, generator=gen)
    x = torch.randn(2, 100, 64, generator=gen)
    t = torch.tensor([801.0, 41.0])
    sd_h = O.round_linear_weights(sd)
    with O.linear_input_rounding(torch.float16):
        cond_w = O.dit_project_cond(sd_h, clip_hidden)
        want = O.dit_forward(sd_h, x, cond_w, t, opt.dit_num_heads)
    cond = m.get_cond(clip_hidden.to(DEV))
    got = m.dit(x.to(DEV), cond, t)
    err = float((got.cpu() - want).abs().max())
    print(f"fp16 DiT forward, latent_size 100: max err vs fp16 emulation {err:.3e}")
    assert err < 5e-3


def test_full_depth_24_dit_32_clip_layers_vs_reference_golden():
    """BASELINE configs[4] at FULL depth: image -> CLIP ViT-H/14 (32 layers) -> proj/norm -> DiT (24
    layers) under CFG 7.5 / DDIM, 3 steps, exact fp32, against tests/golden/dit_full.npz (the reference's own DiT module
    at 24 layers + the 32-layer CLIP restatement, oracle/make_golden.py dit_full).  fp32 vs fp32: the only differences
    are summation orders, amplified by 24 + 32 layers and three guided steps - measured and printed, asserted at 5e-3 on
    O(1) latents."""
    from edgerunner_amd import weights as W
    from edgerunner_amd.models_dit import MDiT
    from edgerunner_amd.options import config_defaults
    path = os.path.join(os.path.dirname(GOLD), "dit_full.npz")
    if not os.path.exists(path):
        pytest.skip("full-depth DiT golden not generated")
    g = dict(np.load(path))
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=2, generate_mode="greedy", cond_mode="point_latent",
                              dit_num_layers=24)
    sd = W.make_dit_state_dict(opt, 0, "perturbed")
    sd.update(W.make_clip_state_dict(32, 0, "perturbed"))
    m = MDiT(opt, DEV, clip_layers=32)
    m.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(int(g["seed"][0]))
    h, w = (int(v) for v in g["image_hw"])
    img = torch.rand(1, 3, h, w, generator=gen)
    noise = torch.randn(1, 2048, 64, generator=gen)
    cond = m.get_cond(img.to(DEV))
    e_cond = float(np.abs(cond[0, [0, 100, 256]].cpu().numpy() - g["cond_rows"]).max())
    r_cond = abs(float(cond.double().sum()) - g["cond_sum"][0]) / g["cond_sum"][1]
    lat = m.run(img.to(DEV), num_inference_steps=int(g["steps"][0]), guidance_scale=7.5, noise=noise.to(DEV))
    rows = g["rows"].tolist()
    e_lat = float(np.abs(lat[0, rows].cpu().numpy() - g["lat_rows"]).max())
    r_lat = abs(float(lat.double().sum()) - g["lat_sum"][0]) / g["lat_sum"][1]
    print(f"full depth (32 CLIP + 24 DiT layers, {int(g['steps'][0])} steps): cond max abs err {e_cond:.3e} (checksum rel {r_cond:.1e}); "
          f"latents max abs err {e_lat:.3e} (checksum rel {r_lat:.1e})")
    assert e_cond < 1e-3 and r_cond < 1e-5
    assert e_lat < 5e-3 and r_lat < 1e-4
    m.close()
