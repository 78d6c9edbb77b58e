#!/usr/bin/env python3
"""bench.py — KD-retrain generator step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one G_Loss_BackProp-equivalent (reference train.py:280-308): student fwd (+ style mixing, fresh
per-layer noise), frozen-D fwd, non-saturating GAN loss, teacher fwd, content-masked L1 distillation, backward
into the student, Adam.  Workload = BASELINE.json configs[1]: 256 px, 70 %-pruned student [154x10,77,77,39,39] +
full teacher, GLOBAL batch 16 (strong scaling: 16/N per GPU), fp32, synthetic seeded weights / latents / mask.
LPIPS and BiSeNet are excluded (weights not obtainable offline) — DESIGN.md §6.

Prints ONE JSON line (rank 0) with the driver's contract fields + `roofline` (dominant hand-written kernel, HIP
events on the launch stream) + `cpu_baseline` (the oracle — a port of the reference CPU path — on host cores)."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GLOBAL_BATCH = 16
SIZE = 256
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
DIRECT_GFLOP_PER_IMG = 301.0  # teacher fwd 45.1 + student fwd 4.12 + bwd 8.24 + D fwd 46.5 + dgrad 46.5 GMAC per image, x 2
PEAK_HBM_GBS = 8000.0


def wino_f4(k_ch, m_ch, B, H, W):
    """This launch runs on the F(4x4,3x3) kernel (conv_wino4.hip) — the LIBRARY's own per-launch decision under its current
    tuning (include/cagc.h cagc_wino_plan = csrc/prep_device.h wino4_for_launch), not a re-derivation."""
    from cagc import _lib
    return _lib.query("cagc_wino_plan", int(B), int(k_ch), int(m_ch), int(H), int(W)) == 4


def wino_macs(k_ch, m_ch, B, H, W):
    return 2.25 if wino_f4(k_ch, m_ch, B, H, W) else 4.0


def up_macs(k_ch, m_ch, B, H, W):
    """MACs per input pixel and channel pair that a stride-2 transposed 3x3 conv EXECUTES: 9 on the direct kernels, 9 * 25 / 36 in the
    Winograd domain of its four output parities (csrc/conv_up25.hip) — the library's own launch decision (cagc_up_plan)."""
    from cagc import _lib
    return 9.0 * _lib.query("cagc_up_plan", int(B), int(k_ch), int(m_ch), int(H), int(W)) / 36.0


def s2_macs(k_ch, m_ch, B, Ho, Wo):
    """MACs per output pixel and channel pair that a 3x3 stride-2 conv EXECUTES: 9 direct, 6.25 on csrc/conv_s2w.hip (cagc_s2_plan)."""
    from cagc import _lib
    return 9.0 * _lib.query("cagc_s2_plan", int(B), int(k_ch), int(m_ch), int(Ho), int(Wo)) / 36.0


def conv_flops(name, a):
    """Algorithmic FLOPs (2 * MACs, the repo's own MAC convention of Util/Calculators.py) of one MFMA launch."""
    if name == "cagc_modconv_fwd":       # (out,x,wp,s,B,Cin,Cout,H,W,k,...)
        B, cin, cout, H, W, k = a[4:10]
        return 2.0 * B * cin * cout * k * k * H * W
    if name == "cagc_modconv_up_fwd":    # (t,x,wp,s,B,Cin,Cout,H,W): 9 MACs per input pixel (6.25 executed in the Winograd domain)
        B, cin, cout, H, W = a[4:9]
        return 2.0 * B * cin * cout * up_macs(cin, cout, B, H, W) * H * W
    if name == "cagc_modconv_dgrad":     # (gx,gs,gz,wp,s,x,B,Cin,Cout,H,W,k)
        B, cin, cout, H, W, k = a[6:12]
        return 2.0 * B * cin * cout * k * k * H * W
    if name == "cagc_modconv_up_dgrad":  # a stride-2 forward conv of the phase-planar gradient: GEMM K = Cout, M = Cin, H x W outputs
        B, cin, cout, H, W = a[6:11]
        from cagc import _lib
        return 2.0 * B * cin * cout * (9.0 * _lib.query("cagc_up_dgrad_plan", int(B), int(cout), int(cin), int(H), int(W)) / 36.0) * H * W
    if name == "cagc_modconv_wgrad":     # (gw,ws,g,x,s,B,Cin,Cout,H,W,k,up,scale)
        B, cin, cout, H, W, k = a[5:11]
        return 2.0 * B * cin * cout * k * k * H * W
    if name == "cagc_modconv_wgrad_demod":   # (gw,ws,g,x,s,gwsq,weight,B,Cin,Cout,H,W,k,up,scale)
        B, cin, cout, H, W, k = a[7:13]
        return 2.0 * B * cin * cout * k * k * H * W
    if name == "cagc_wino_conv3x3":      # (out,x,up,s,B,Cin,Cout,H,W,...): Winograd — count the flops the MFMA pipe EXECUTES
        B, cin, cout, H, W = a[4:9]      # (F(2x2,3x3): 16 GEMMs over H/2*W/2 tiles = 4 MACs per output pixel and channel pair;
        return 2.0 * B * cin * cout * wino_macs(cin, cout, B, H, W) * H * W   # F(4x4,3x3): 36 over H/4*W/4 = 2.25), not the direct conv's 9
    if name == "cagc_wino_conv3x3_act_dgrad":   # (gx,gout,act_out,up,residual,B,Cin,Cout,H,W,...)
        B, cin, cout, H, W = a[5:10]
        return 2.0 * B * cin * cout * wino_macs(cout, cin, B, H, W) * H * W     # data gradient: GEMM K = Cout, M = Cin
    if name == "cagc_conv3x3s2_fwd":     # (out,x,wp,B,Cin,Cout,Hin,Win,pitch)
        B, cin, cout, hin, win = a[3:8]
        ho, wo = (hin - 3) // 2 + 1, (win - 3) // 2 + 1
        return 2.0 * B * cin * cout * s2_macs(cin, cout, B, ho, wo) * ho * wo
    if name == "cagc_conv3x3s2_dgrad":   # same arguments; a transposed conv of the (ho x wo) gradient: GEMM K = Cout, M = Cin
        B, cin, cout, hin, win = a[3:8]
        ho, wo = (hin - 3) // 2 + 1, (win - 3) // 2 + 1
        return 2.0 * B * cin * cout * up_macs(cout, cin, B, ho, wo) * ho * wo
    if name == "cagc_conv3x3s2_act_fwd":     # (out,x,wp,bias,B,Cin,Cout,Hin,Win,pitch,...)
        B, cin, cout, hin, win = a[4:9]
        ho, wo = (hin - 3) // 2 + 1, (win - 3) // 2 + 1
        return 2.0 * B * cin * cout * s2_macs(cin, cout, B, ho, wo) * ho * wo
    if name == "cagc_gemm1x1":           # (out,x,ap,residual,B,K,M,P,alpha,beta): the ResBlock skip's 1x1 conv, one GEMM per image
        B, K, M, P = a[4:8]
        return 2.0 * B * K * M * P
    return 0.0


def stream_bytes(name, a):
    """Algorithmic HBM bytes (each element read / written once) of one launch of the streaming entry points."""
    if name == "cagc_fused_bias_act_bwd":     # (gx,gbias,gout,out,outer,C,inner,...): read gout + out, write gx
        return 12.0 * a[4] * a[5] * a[6]
    if name == "cagc_fused_bias_act_fwd":     # (out,x,bias,outer,C,inner,...)
        return 8.0 * a[3] * a[4] * a[5]
    if name == "cagc_fir4x4_pitched":         # (out,x,k,planes,in_h,in_w,in_pitch,out_h,out_w,out_pitch,...)
        return 4.0 * a[3] * (a[4] * a[5] + a[7] * a[8])
    if name == "cagc_upfirdn2d":              # (out,x,k,planes,in_h,in_w,out_h,out_w,...)
        return 4.0 * a[3] * (a[4] * a[5] + a[6] * a[7])
    if name == "cagc_fir4x4_up2_acc":         # (out,x,k,acc,planes,in_h,in_w,out_h,out_w): read x + acc, write out
        return 4.0 * a[4] * (a[5] * a[6] + 2 * a[7] * a[8])
    if name in ("cagc_fromrgb_act_dgrad",):   # (gx,gout,act_out,w,B,C,HW,...): read gout + out, write 3 channels
        return 4.0 * a[4] * a[6] * (2 * a[5] + 3)
    if name == "cagc_fromrgb_fwd":            # (out,x,w,bias,B,C,HW,...)
        return 4.0 * a[4] * a[6] * (a[5] + 3)
    if name == "cagc_blur_up_fwd":            # (out,t,fir,d,noise,nb,nw,bias,B,C,H,W,...): 4 phase planes in, 2Hx2W out
        B, C, H, W = a[8:12]
        return 4.0 * B * C * (4 * (H + 1) * (W + 1) + 4 * H * W)
    if name == "cagc_blur_up_bwd":            # (gt,gz,fir,B,C,H,W): 2Hx2W gradient in, 4 phase planes out
        B, C, H, W = a[3:7]
        return 4.0 * B * C * (4 * (H + 1) * (W + 1) + 4 * H * W)
    if name == "cagc_torgb_fwd":              # (out,x,w,s,bias,skip,fir,B,C,H,W,scale): read x (+ skip at half size), write 3 channels
        B, C, H, W = a[7:11]
        return 4.0 * B * H * W * (C + 3 + (0.75 if a[5] is not None else 0.0))
    if name == "cagc_torgb_bwd":              # (gx,gws,g,x,w,s,B,C,H,W,scale): read g (3 ch) + x, write gx
        B, C, H, W = a[6:10]
        return 4.0 * B * H * W * (2 * C + 3)
    if name == "cagc_torgb_bwd_finish":       # (gw,gs,gbias,gws,s,w,B,C,scale): [B,3,C+1] sums in, [3,C] + [B,C] out
        B, C = a[6:8]
        return 4.0 * (B * 3 * (C + 1) + 2 * B * C + 6 * C)
    if name == "cagc_styled_act_bwd":         # (gz,red,gout,out,d,noise,nb,B,C,HW,...): read gout + out (+ noise), write gz
        nb, B, C, HW = a[6:10]
        return 4.0 * HW * (3 * B * C + nb)
    if name == "cagc_scale_reduce":           # (gx,x,s,gs,B,C,HW): read gx + x, write gx
        return 12.0 * a[4] * a[5] * a[6]
    if name == "cagc_maplin_fwd":             # (y,x,w,b,R,in,out,...): the weight matrix is the traffic (R <= 256 rows)
        R, D, O = a[4:7]
        return 4.0 * (D * O + R * D + R * O)
    if name == "cagc_maplin_bwd":             # (gx,gw,gb,gy,y,x,w,R,in,out,...): read w, write gw
        R, D, O = a[7:10]
        return 4.0 * (2 * D * O + 2 * R * D + 2 * R * O)
    if name == "cagc_modbank_fwd":            # (out,latent,wptrs,bptrs,meta,L,Ctot,B,n_latent,style_dim,scale): every modulation matrix once
        L, Ctot, B, nl, D = a[5:10]
        return 4.0 * (Ctot * D + B * nl * D + B * Ctot)
    if name == "cagc_modbank_bwd":            # (gw,gb,g_latent,gs,latent,wptrs,meta,L,Ctot,B,n_latent,style_dim,scale)
        L, Ctot, B, nl, D = a[7:12]
        return 4.0 * (2 * Ctot * D + 2 * B * nl * D + B * Ctot)
    if name == "cagc_styled_bwd_tail":        # (gbias,gnw,gs,gwsq,red,bias,nw,d,s,wsq,B,Cin,Cout,has_noise): wsq in, gwsq out
        B, cin, cout = a[10:13]
        return 4.0 * (2 * cin * cout + 5 * B * cout + 2 * B * cin)
    if name == "cagc_gan_kd_loss_tail":       # (out3,gs,gpred,pred,P,t,s,mask,B,C,HW,...): read teacher + student + mask, write gs
        B, C, HW = a[8:11]
        return 4.0 * B * HW * (3 * C + 1)
    if name == "cagc_pixelnorm_fwd":          # (y,x,rows,dim)
        return 8.0 * a[2] * a[3]
    if name == "cagc_add_scale":              # (out,a,b,n,scale)
        return 12.0 * a[3]
    if name == "cagc_mix_latent_fwd":         # (latent,w0,w1,inject,B,n_latent,D)
        return 4.0 * a[4] * (a[5] + 2) * a[6]
    if name == "cagc_mix_latent_bwd":         # (gw0,gw1,g,inject,B,n_latent,D)
        return 4.0 * a[4] * (a[5] + 2) * a[6]
    if name == "cagc_modconv_prep_bank":      # (jobs,njobs): weights in; packed fwd / bwd operands, wsq and Winograd-domain weights out
        tot = 0.0
        for j in a[0][:a[1]]:
            w = float(j.Cout) * j.Cin * j.ksize * j.ksize
            tot += 4.0 * w * (1 + (1 if j.wp_fwd else 0) + (1 if j.wp_bwd else 0)) + 4.0 * j.Cout * j.Cin * (
                (1 if j.wsq else 0) + (16 if j.up_fwd else 0) + (16 if j.up_bwd else 0))
        return tot
    if name == "cagc_demod_bank":             # (jobs,njobs,B): wsq [Cout,Cin] + s [B,Cin] in, d [B,Cout] out, per layer
        return sum(4.0 * (j.Cout * j.Cin + a[2] * (j.Cin + j.Cout)) for j in a[0][:a[1]])
    return 0.0


# dominant-entry-point -> device symbol (for the PMC traffic lookup) and MFMA instruction
KERNEL_OF = {"cagc_wino_conv3x3[k_wino<4, false>]": "k_wino<4, false, false", "cagc_wino_conv3x3_act_dgrad[k_wino<4, true>]": "k_wino<4, true, false",
             "cagc_wino_conv3x3[k_wino<3, false>]": "k_wino<3, false, false",
             "cagc_wino_conv3x3[k_wino<4, false, NH3>]": "k_wino<4, false, false, 3>", "cagc_wino_conv3x3_act_dgrad[k_wino<4, true, NH3>]": "k_wino<4, true, false, 3>",
             "cagc_wino_conv3x3[k_wino<4, false, NH2>]": "k_wino<4, false, false, 2>", "cagc_wino_conv3x3[k_wino<4, false, NH1>]": "k_wino<4, false, false, 1>",
             "cagc_wino_conv3x3[k_wino<3, false, NH1>]": "k_wino<3, false, false, 1>", "cagc_wino_conv3x3[k_wino<3, false, NH2>]": "k_wino<3, false, false, 2>",
             "cagc_wino_conv3x3_act_dgrad[k_wino<4, true, NH2>]": "k_wino<4, true, false, 2>", "cagc_wino_conv3x3_act_dgrad[k_wino<4, true, NH1>]": "k_wino<4, true, false, 1>",
             "cagc_wino_conv3x3[k_wino4<false>]": "k_wino4<false,", "cagc_wino_conv3x3_act_dgrad[k_wino4<true>]": "k_wino4<true,",   # both SCALE variants
             "cagc_modconv_fwd": "k_conv_rd<4, true, true, false>",
             "cagc_modconv_up_fwd": "k_conv_up25<true, 0",
             "cagc_conv3x3s2_fwd": "k_conv_rd<8, false, false, false>", "cagc_conv3x3s2_act_fwd": "k_conv_s2w<1, 4, false>", "cagc_conv3x3s2_dgrad": "k_conv_up25<false, 1",
             "cagc_modconv_dgrad": "k_conv_rd<5, true, false, true>",
             "cagc_modconv_up_dgrad": "k_conv_s2w<2, 5, true>", "cagc_modconv_wgrad": "k_wgrad_rd<4, 1, false, 9>",
             "cagc_modconv_wgrad_demod": "k_wgrad_rd<3, 1, false, 9>"}


def pmc_traffic(symbol):
    """HBM-side bytes per launch of `symbol` from the committed rocprofv3 PMC passes (profiles/r0N_pmc_*.md, newest round; separate
    --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same bench, counters in KB).  gfx950 correction
    (MI355X_MICROARCH.md §HBM, re-calibrated here on the bias+act stream kernel whose byte count is known exactly:
    FETCH_SIZE reads 0.50x, WRITE_SIZE 1.00x of the true bytes): bytes = 2*FETCH + WRITE."""
    if SIZE != 256:
        return None, "the committed PMC passes profile the 256 px workload"
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = next((q for q in (os.path.join(ROOT, "profiles", f"{r}_pmc_{c}.md") for r in ("r06", "r05", "r04", "r03")) if os.path.exists(q)), None)
        if path is None:
            return None, "no committed PMC summary"
        for line in open(path):
            cols = [x.strip() for x in line.split("|")]
            if len(cols) > 6 and symbol in cols[1] and cols[2] == c:     # a kernel family (symbol prefix) may match several rows
                acc = vals.setdefault(c + "_acc", [0.0, 0.0])
                acc[0] += float(cols[4]); acc[1] += float(cols[3])
                vals[c] = acc[0] * 1024.0 / acc[1]
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, "kernel not found in PMC summary"
    return int(2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]), (
        f"avg HBM bytes/launch of {symbol} from the committed rocprofv3 --pmc passes (profiles/r0N_pmc_*.md, newest round): "
        f"2x FETCH_SIZE + WRITE_SIZE; MFMA-bound kernel, re-reads of the input tiles by the channel tiles are served "
        f"by L2/MALL")


class KernelTimer:
    """HIP-event timing of every libcagc entry point, on the stream the kernels are launched on."""

    def __init__(self, lib_mod):
        self.lib_mod = lib_mod
        self.orig = lib_mod.call
        self.records = []

    def __enter__(self):
        def timed(name, *args):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            s.record(st)
            self.orig(name, *args)
            e.record(st)
            key = name
            if name in ("cagc_wino_conv3x3", "cagc_wino_conv3x3_act_dgrad") and (
                    wino_f4(args[7], args[6], args[5], args[8], args[9]) if name.endswith("act_dgrad") else wino_f4(args[5], args[6], args[4], args[7], args[8])):
                key = f"{name}[k_wino4<{'true' if name.endswith('act_dgrad') else 'false'}>]"       # F(4x4,3x3): conv_wino4.hip
            elif name in ("cagc_wino_conv3x3", "cagc_wino_conv3x3_act_dgrad"):   # one record family per device kernel, so the
                gated = name.endswith("act_dgrad")                              # average launch duration is comparable with the
                nblk = -(-(args[6]) // 16)                                      # rocprofv3 per-symbol summary (conv_wino.hip wino_mb)
                mb = nblk if nblk <= 3 else (3 if -(-nblk // 3) * 3 < -(-nblk // 4) * 4 else 4)
                # 4-wave (NH 1) or 8-wave (NH 2) workgroups: conv_wino.hip wino_nh()
                Bq, K, Hq, Wq = (args[5], args[7], args[8], args[9]) if gated else (args[4], args[5], args[7], args[8])
                wgs2 = Bq * (Wq // 32) * (Hq // 8) * -(-args[6] // (mb * 16))
                nh = 1 if (-(-K // 16) * 16 <= 128 or wgs2 < 1024) else (3 if (mb == 4 and args[6] % 128 == 0) else 2)   # 3: wide shape
                key = f"{name}[k_wino<{mb}, {'true' if gated else 'false'}, NH{nh}>]"
            self.records.append((key, s, e, conv_flops(name, args), stream_bytes(name, args)))
        self.lib_mod.call = timed
        return self

    def __exit__(self, *a):
        self.lib_mod.call = self.orig

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, s, e, fl, by in self.records:
            d = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e)
            d[2] += fl
            d[3] += by
        return agg


def cpu_baseline(steps=3, batch=16, budget_s=120.0, threads=None):
    """The oracle (port of the reference CPU path: grouped convs on per-sample modulated weights, composed
    upfirdn2d / leaky_relu) timed on the host cores, SURVEY §8-d: one warm-up step (batch 2: oneDNN primitive creation)
    then up to `steps` timed KD generator steps at batch `batch` of the same 256 px workload; the median is reported.
    Bounded: no further step is started once `budget_s` seconds of timed work have elapsed (the reference's own CPU path
    took 57 s per bs-16 step on 8 cores)."""
    from cagc import kd
    from oracle import ref_kd
    # intra-op threads: all cores up to 32 (the grouped-conv CPU kernels stop scaling — and oversubscribe badly —
    # beyond that; measured 5x slower with 256 threads on a 2x64-core host)
    cores = threads if threads else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    student, teacher, disc = kd.build_synthetic_workload(SIZE, "cpu", seed=0)
    ssd = {k: v.detach() for k, v in student.state_dict().items()}
    names = [n for n, _ in student.named_parameters()]
    tsd, dsd = dict(teacher.state_dict()), dict(disc.state_dict())

    def one(bs):
        g = torch.Generator().manual_seed(99)
        zs = [torch.randn(bs, 512, generator=g), torch.randn(bs, 512, generator=g)]
        mask = kd.ellipse_mask(bs, SIZE, "cpu")
        leaves = {k: ssd[k].clone().requires_grad_(True) for k in names}
        sd = dict(ssd)
        sd.update(leaves)
        t0 = time.perf_counter()
        g_loss, kd_l1, _ = ref_kd.kd_generator_losses_ref(sd, tsd, dsd, zs, 5, mask)
        grads = torch.autograd.grad(g_loss + kd_l1, [leaves[k] for k in names], allow_unused=True)
        ref_kd.adam_step_ref({k: ssd[k] for k in names},
                             {k: (g if g is not None else torch.zeros_like(ssd[k])) for k, g in zip(names, grads)}, {},
                             0.0016, (0.0, 0.99 ** 0.8))
        return time.perf_counter() - t0
    one(2)   # warm-up
    times = []
    while len(times) < steps and sum(times) < budget_s:
        times.append(one(batch))
    med = sorted(times)[len(times) // 2]
    return {"value": round(batch / med, 4), "unit": "images/s", "cores": cores, "host_logical_cpus": os.cpu_count(), "kind": "port",
            "cores_note": "intra-op threads = min(os.cpu_count(), 32): the grouped-conv CPU kernels stop scaling there and oversubscribe beyond it "
                          "(measured 5x SLOWER with all 256 hardware threads of a 2 x EPYC 9575F host: 0.0134 img/s; --cpu-all-cores times that too)",
            "sample": f"{len(times)} timed KD generator step(s) at batch {batch} of the 256px bs16 workload after a batch-2 warm-up; "
                      f"median {med:.1f} s/step (all: {', '.join(f'{t:.1f}' for t in times)} s)",
            "cpu_model": _cpu_model()}


def configs0_cpu_forward(runs=3, batch=4, threads=None):
    """BASELINE configs[0] through the PRODUCT's own CPU path (the composed-PyTorch branches of cagc/op/fused_act.py and
    cagc/op/upfirdn2d.py — what the reference's ops do on CPU tensors, op/fused_act.py:105-116, op/upfirdn2d.py:146-149): the full
    256 px Generator forward on random latents at batch 4, plumbing only (no GPU, no oracle).  BASELINE.md §2 has the reference at
    2.3-2.7 s on 8 cores."""
    from cagc import model as M
    cores = threads if threads else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    gen = M.Generator(256, 512, 8).eval()
    z = [torch.randn(batch, 512)]
    times = []
    with torch.no_grad():
        gen(z)                                   # warm-up (oneDNN primitive creation)
        for _ in range(runs):
            t0 = time.perf_counter()
            img = gen(z)
            times.append(time.perf_counter() - t0)
    assert tuple(img.shape) == (batch, 3, 256, 256) and bool(torch.isfinite(img).all())
    med = sorted(times)[len(times) // 2]
    return {"value": round(batch / med, 3), "unit": "images/s", "seconds_per_forward": round(med, 3), "cores": cores, "batch": batch,
            "what": "configs[0]: 256px StyleGAN2 Generator forward, random latents, bs=4, the product's CPU path (torch-native upfirdn2d / fused_act "
                    "branches; no HIP library, no oracle)"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _graph_replay_clock(kd, student, teacher, disc, bs, mask, dev, rng, gen, n=10):
    """Shader clock (MHz) the MFMA kernels of the HIP-graph-replayed KD step see — the launch mode of the timed region at N = 1.  The probe
    pointer is read at launch time, so it is set BEFORE the capture (include/cagc.h cagc_set_clock_probe) and every replay carries it: every
    64th workgroup of every k_wino4 / k_conv_rd launch adds its own clock.  Runs after the timed region, on a copy of the student."""
    import copy, ctypes
    from cagc import _lib
    acc = torch.zeros(2, device=dev)
    lib = _lib.load()
    lib.cagc_set_clock_probe(ctypes.c_void_p(acc.data_ptr()))
    try:
        st2 = kd.GraphedKDStep(copy.deepcopy(student), teacher, disc, bs, mask, random_noise=True, world_size=1)
    finally:
        lib.cagc_set_clock_probe(None)          # launches outside the captured graphs stop sampling
    for _ in range(3):
        st2.sample_and_step(bs, mask, rng, gen)
    torch.cuda.synchronize()
    acc.zero_()
    t1 = time.perf_counter()
    for _ in range(n):
        st2.sample_and_step(bs, mask, rng, gen)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / n * 1e3
    out = {"mhz": round(float(acc[0] / acc[1])) if float(acc[1]) > 0 else None, "samples_per_step": int(float(acc[1]) / n),
           "ms_per_step_with_probe": round(ms, 3),
           "what": "mean over every 64th workgroup of every F(4x4) Winograd and register-direct conv launch of the replayed step"}
    del st2
    return out


def _time_mode(kd, cd, student, teacher, disc, bs, mask, world, dev, rng, gen, mode, n=8):
    """ms/step of `n` steps of one launch mode on a COPY of the student (untimed calibration / proxy runs)."""
    import copy
    s2 = copy.deepcopy(student)
    if mode == "graph":
        st2 = kd.GraphedKDStep(s2, teacher, disc, bs, mask, random_noise=True, world_size=world)
    else:
        st2 = kd.KDStep(cd.wrap_student(s2, dev), teacher, disc)
    for _ in range(3):
        st2.sample_and_step(bs, mask, rng, gen)
    cd.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        st2.sample_and_step(bs, mask, rng, gen)
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    del st2, s2
    return t.item() / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--full-iteration", action="store_true", help="(default at N=1; kept for compatibility)")
    ap.add_argument("--no-full-iteration", action="store_true",
                    help="skip the secondary leg that times 16 whole training iterations (D step, R1, path-length, EMA)")
    ap.add_argument("--sweep", type=int, default=10, help="time N bs-64 batches of the prune.py saliency sweep (config 5); 0 = skip")
    ap.add_argument("--size", type=int, default=256, choices=(256, 1024),
                    help="256: BASELINE configs[1] (the headline).  1024: configs[3] — the 1024 px pruned student [..,20,20,10,10] + "
                         "full 1024 px teacher + Discriminator(1024), global batch 16 over 4 GPUs; at --gpus 1 ONE rank's share "
                         "(per-GPU batch 4) is timed, the secondary legs are skipped")
    ap.add_argument("--local-batch", type=int, default=0, help="per-GPU batch override (default: global batch 16 / world size; "
                                                               "--size 1024 at --gpus 1: 4)")
    ap.add_argument("--no-config3", action="store_true", help="skip the compact configs[3] (1024 px, per-GPU batch 4) leg")
    ap.add_argument("--no-sweep-clock", action="store_true", help="do not run the shader-clock probe during the saliency-sweep leg")
    ap.add_argument("--no-proxy", action="store_true", help="skip the strong-scaling proxy table (per-GPU batch 8/4/2 on this GPU)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed bs-16 steps of the CPU baseline (SURVEY 8-d: 3)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baseline with os.cpu_count() threads (~5 min)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches + DDP instead of HIP-graph replay")
    ap.add_argument("--graph", action="store_true", help="force HIP-graph replay (default: calibrate — a few untimed steps "
                                                         "of each mode on copies of the models, keep the faster)")
    args = ap.parse_args()

    global SIZE
    SIZE = args.size
    from cagc import _lib, distributed as cd, kd
    rank, world, local = cd.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    _lib.load()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # MIOpen exhaustive find for the discriminator's stock convs is opt-in (CAGC_MIOPEN_BENCHMARK=1)
    torch.backends.cudnn.benchmark = os.environ.get("CAGC_MIOPEN_BENCHMARK", "0") == "1"
    assert GLOBAL_BATCH % world == 0
    bs = GLOBAL_BATCH // world
    if os.environ.get("CAGC_BENCH_LOCAL_BS"):   # experiment knob: per-GPU batch of an N-GPU run, on one GPU (value then = bs*K/t)
        bs = int(os.environ["CAGC_BENCH_LOCAL_BS"])
    share = None
    if args.local_batch:
        bs = args.local_batch
    elif SIZE == 1024 and world == 1:
        bs = 4                 # configs[3] is a 4-GPU run of global batch 16: one rank's share
    if bs * world != GLOBAL_BATCH:
        share = f"one GPU at per-GPU batch {bs} = the share of one rank of a {GLOBAL_BATCH // bs}-GPU run of global batch {GLOBAL_BATCH}"
    if SIZE != 256:            # the secondary legs belong to the 256 px headline configuration
        args.no_full_iteration, args.no_proxy, args.sweep, args.no_cpu_baseline, args.no_config3 = True, True, 0, True, True

    student, teacher, disc = kd.build_synthetic_workload(SIZE, dev, seed=0)
    n_params = sum(p.numel() for p in student.parameters())
    mask = kd.ellipse_mask(bs, SIZE, dev)
    rng = random.Random(rank)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    torch.cuda.manual_seed(1234 + rank)
    mode = "graph"
    step = None
    calib = None
    if not args.no_graph and not args.graph:
        # Launch-mode calibration (untimed, before the warm-up): the same step either replayed as HIP graphs or launched
        # eagerly (teacher on its own stream, DDP buckets overlapping backward).  Which one wins depends on the per-GPU
        # batch and on the host CPU; both compute the same thing (tests/test_gpu_parity.py::test_graphed_kd_step...).
        # Every rank must take the same branch (the modes issue different collectives): failures are all-reduced.
        tg, ok = {}, 1.0
        for m in ("graph", "eager"):
            try:
                tg[m] = _time_mode(kd, cd, student, teacher, disc, bs, mask, world, dev, rng, gen, m)
            except Exception as e:  # noqa: BLE001 — calibration is an optimisation only
                print(f"[bench] launch-mode calibration of '{m}' failed ({type(e).__name__}: {e})", file=sys.stderr)
                ok = 0.0
                break
        okt = torch.tensor([ok], device=dev)
        if world > 1:
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if okt.item() > 0:
            calib = {m: round(v, 3) for m, v in tg.items()}
            args.no_graph = calib["eager"] < calib["graph"]      # identical on every rank (MAX-reduced times)
        torch.cuda.empty_cache()
    if not args.no_graph:
        ok = 1.0
        try:     # HIP-graph replay of the step; gradients all-reduced as one flat RCCL collective between graphs
            step = kd.GraphedKDStep(student, teacher, disc, bs, mask, random_noise=True, world_size=world)
        except Exception as e:  # noqa: BLE001 — capture is an optimisation, never a correctness requirement
            print(f"[bench] HIP-graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            ok = 0.0
        okt = torch.tensor([ok], device=dev)
        if world > 1:
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if okt.item() == 0:   # any rank failed: ALL ranks fall back to the eager mode on fresh models
            step = None
            torch.cuda.synchronize()
            student, teacher, disc = kd.build_synthetic_workload(SIZE, dev, seed=0)
    if step is None:
        mode = "eager"
        ddp_student = cd.wrap_student(student, dev)
        step = kd.KDStep(ddp_student, teacher, disc)

    def run(n, marks=None):
        for _ in range(n):
            step.sample_and_step(bs, mask, rng, gen)
            if marks is not None:     # per-step completion marks for the median (no host sync inside the timed region)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)

    run(args.warmup)
    cd.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    marks = []
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps, marks)
    torch.cuda.synchronize()
    cd.barrier()
    dt = time.perf_counter() - t0
    stamps = [ev0.elapsed_time(e) for e in marks]
    per_step = sorted(b - a for a, b in zip([0.0] + stamps[:-1], stamps))
    median_ms = per_step[len(per_step) // 2] if per_step else None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = bs * world * args.steps / dt

    # The gradient all-reduce by itself (N > 1): every bucket of the flat gradient reduced on its own, HIP events on the collective's stream,
    # median of 5 — what the step's timeline has to hide (scripts/scale.sh prints it next to the step time).
    bucket_ms = bucket_bytes = None
    if world > 1:
        flat = getattr(step, "flat_grad", None)
        spans = [(lo, hi) for lo, hi, _, _ in getattr(step, "_buckets", [])] if flat is not None else []
        if flat is None:      # eager DDP: one flat buffer of the gradient's size, cut like DDP's 4 MB buckets
            flat = torch.zeros(n_params, device=dev)
            per = (4 << 20) // 4
            spans = [(i, min(i + per, n_params)) for i in range(0, n_params, per)]
        probe = flat.clone()
        bucket_ms, bucket_bytes = [], []
        for lo, hi in spans:
            ts = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                cd.barrier()
                e0.record()
                dist.all_reduce(probe[lo:hi])
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            bucket_ms.append(round(sorted(ts[1:])[2], 4))
            bucket_bytes.append(4 * (hi - lo))
        del probe

    roof = None
    if not args.no_roofline:
        # per-kernel HIP-event timing needs individual launches: same models, eager launches (no graph), no DDP
        prof_step = step if mode == "eager" else kd.KDStep(student, teacher, disc)
        from cagc.op import modconv as _mc
        overlap_saved, kd.OVERLAP_TEACHER = kd.OVERLAP_TEACHER, False   # one stream: kernels are timed in isolation
        side_saved, _mc._SIDE_LIMIT = _mc._SIDE_LIMIT, 0                # (also the weight gradients / skip GEMMs of the side stream)
        fork_saved, _mc.FORK_TORGB = _mc.FORK_TORGB, False              # (and the student's ToRGB chain)
        for _ in range(2):
            prof_step.sample_and_step(bs, mask, rng, None)
        import ctypes as _ct
        clk_acc = torch.zeros(2, device=dev)     # include/cagc.h cagc_set_clock_probe: shader clock seen inside the F(4x4) launches
        _lib.load().cagc_set_clock_probe(_ct.c_void_p(clk_acc.data_ptr()))
        with KernelTimer(_lib) as kt:
            for _ in range(3):
                prof_step.sample_and_step(bs, mask, rng, None)
        agg = kt.summary()
        clk_mhz = float(clk_acc[0] / clk_acc[1]) if float(clk_acc[1]) > 0 else None
        # the dominant kernel's OWN clock: the probe restricted to the F(4x4) Winograd launches (cagc_set_tuning("clock_probe_family", 1)).
        # The all-kernel mean above mixes in the register-direct convs, which hold 2.3-2.4 GHz; k_wino4 itself runs lower.
        clk_acc.zero_()
        with _lib.tuning(clock_probe_family=1):
            for _ in range(2):
                prof_step.sample_and_step(bs, mask, rng, None)
            torch.cuda.synchronize()
        _lib.load().cagc_set_clock_probe(None)
        clk_w4 = float(clk_acc[0] / clk_acc[1]) if float(clk_acc[1]) > 0 else None
        kd.OVERLAP_TEACHER = overlap_saved
        _mc._SIDE_LIMIT = side_saved
        _mc.FORK_TORGB = fork_saved
        if rank == 0:
            mfma = {k: v for k, v in agg.items() if v[2] > 0}
            # the Winograd kernel runs as 4-wave (NH1) or 8-wave (NH2) workgroups of the SAME source kernel, chosen per launch
            # (conv_wino.hip wino_nh): dominance and `achieved` are taken over the kernel (both tile heights), the per-symbol
            # rows that rocprofv3 lists are in `variants`
            import re as _re
            fam_of = lambda k: _re.sub(r", NH\d", "", k)
            fams = {}
            for k, v in mfma.items():
                f = fams.setdefault(fam_of(k), [0, 0.0, 0.0, 0.0])
                for i in range(4):
                    f[i] += v[i]
            name, (cnt, tot_ms, flops, _) = max(fams.items(), key=lambda kv: kv[1][1])
            ach = flops / (tot_ms * 1e-3) / 1e12
            sym = KERNEL_OF.get(name, name)
            traffic, traffic_note = pmc_traffic(sym)
            variants = {k: {"device_symbol": KERNEL_OF.get(k, k), "launches_per_step": v[0] // 3, "avg_launch_ms": round(v[1] / v[0], 4),
                            "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in mfma.items() if fam_of(k) == name and k != name}
            roof = {"bound": "mfma", "kernel": f"{name} ({sym}, v_mfma_f32_16x16x4_f32)", "variants": variants,
                    "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "launches_per_step": cnt // 3, "avg_launch_ms": round(tot_ms / cnt, 4),
                    "flops_per_launch_avg": flops / cnt,
                    # `peak` is the guide's figure at 2.4 GHz; under real operands the F(4x4) kernel runs at the board's power limit (back to
                    # back: 2.0 GHz; 2.35 GHz on all-zero operands, same instruction stream: DESIGN.md §5).  The probe averages every 64th
                    # workgroup of every probed launch (F(4x4) and register-direct convs) of these eagerly launched steps:
                    "leg": "eager launches on ONE stream (HIP events need individual launches; teacher / weight-gradient / ToRGB side streams off) — the "
                           "timed region replays the same step from a HIP graph with those streams on",
                    "shader_clock_mhz": None if clk_mhz is None else round(clk_mhz),
                    "shader_clock_note": "mean over every 64th workgroup of every F(4x4) Winograd and register-direct / stream-K conv launch of these eagerly launched steps (cagc_set_clock_probe)",
                    "dominant_kernel_clock_mhz": None if clk_w4 is None else round(clk_w4),
                    "dominant_kernel_clock_note": "the same probe restricted to the k_wino4 launches (cagc_set_tuning('clock_probe_family', 1)): the clock the dominant kernel itself runs at inside the step",
                    "frac_at_measured_clock": None if (clk_w4 or clk_mhz) is None else round(ach / (PEAK_F32_MFMA_TFLOPS * (clk_w4 or clk_mhz) / 2400.0), 4),
                    "frac_at_measured_clock_note": "achieved / (peak x dominant_kernel_clock / 2400 MHz): the share of the matrix pipe's cycles at the clock the kernel ran at",
                    "achieved_direct_conv_equivalent": round(ach * (4.0 if "k_wino4" in name else (2.25 if name.startswith("cagc_wino_conv3x3") else 1.0)), 2),
                    "flops_note": ("MFMA flops executed (Winograd F(4x4,3x3): 2.25 MACs/output/channel-pair; its direct-conv equivalent "
                                   "rate is 4x 'achieved')" if "k_wino4" in name else
                                   "MFMA flops executed (Winograd F(2x2,3x3): 4 MACs/output/channel-pair; its direct-conv equivalent "
                                   "rate is 2.25x 'achieved')" if name.startswith("cagc_wino_conv3x3") else "2*MACs of the conv"),
                    "all_mfma_entry_points": {k: {"ms_per_step": round(v[1] / 3, 3), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                                              for k, v in sorted(mfma.items(), key=lambda kv: -kv[1][1])},
                    "cagc_kernel_ms_per_step": {k: round(v[1] / 3, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
            # step-level figures: what the TFLOP/s in this line mean.  `executed` = MFMA flops the kernels issue (Winograd F(4x4): 2.25,
            # F(2x2): 4, parity-domain stride-2: 6.25 of the direct conv's 9 MACs per output and channel pair); `direct_equivalent` = the
            # analytic count of SURVEY 8-d (301 GFLOP per image: what a direct-convolution implementation would execute) — it may exceed
            # the 157.3 TFLOP/s fp32-MFMA peak and is NOT a utilisation figure.
            k_ms = sum(v[1] for v in agg.values()) / 3
            k_attr = sum(v[1] for v in agg.values() if v[2] > 0 or v[3] > 0) / 3
            exec_flops = sum(v[2] for v in agg.values()) / 3
            roof["step_kernel_ms_one_stream"] = round(k_ms, 3)
            roof["frac_of_step_attributed"] = round(k_attr / k_ms, 4) if k_ms > 0 else None
            roof["step_executed_tflops"] = round(exec_flops / (ms * 1e-3) / 1e12, 1)
            roof["step_direct_equivalent_tflops"] = round(DIRECT_GFLOP_PER_IMG * 1e9 * bs / (ms * 1e-3) / 1e12, 1) if SIZE == 256 else None
            roof["step_tflops_note"] = ("per rank, over the TIMED ms_per_step: executed = MFMA flops issued by the libcagc launches of one step; direct_equivalent = "
                                        "301 GFLOP/img (SURVEY 8-d, every conv counted as a direct convolution) — not a utilisation figure, may exceed the peak")
            roof["unattributed_entry_points"] = {k: round(v[1] / 3, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]) if v[2] == 0 and v[3] == 0}
            hbm = {k: v for k, v in agg.items() if v[3] > 0}
            if hbm:   # secondary: the HBM-bound streaming kernels against the 8 TB/s HBM3E peak (SURVEY 8-d "report both")
                big = {}     # the largest launch of each entry point on its own (small launches are latency-bound)
                for name, s_ev, e_ev, _, by in kt.records:
                    if by > 0 and by >= big.get(name, (0, 0.0))[0]:
                        ms_l = s_ev.elapsed_time(e_ev)
                        if by > big.get(name, (0, 0.0))[0] or ms_l < big[name][1]:
                            big[name] = (by, ms_l)
                roof["hbm_bound_entry_points"] = {
                    k: {"ms_per_step": round(v[1] / 3, 3), "achieved_GBps": round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                        "frac_of_8TBps": round(v[3] / (v[1] * 1e-3) / 8e12, 3),
                        "largest_launch_MB": round(big[k][0] / 1e6, 1),
                        "largest_launch_GBps": round(big[k][0] / (big[k][1] * 1e-3) / 1e9, 1)}
                    for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])}
    full = None
    if world == 1 and not args.no_full_iteration:
        # secondary figure: the WHOLE training iteration of train.py:371-398 — D step + G/KD step + lazy
        # R1 (every 16) + lazy path-length reg (every 4) + EMA — eagerly launched, 16 iterations = one full lazy-reg
        # period.  Comparable in kind to the reference's README.md:108-115 wall-time figure (15.3 img/s on 2xV100,
        # which also included BiSeNet, LPIPS, data loading and FID).
        import copy
        g_ema = copy.deepcopy(student)
        it = kd.TrainIteration(student, teacher, disc, g_ema=g_ema)
        real = torch.rand(bs, 3, SIZE, SIZE, device=dev) * 2 - 1
        for i in range(1, 4):
            it.iteration(i, real, mask, rng, None)
        it.iteration(0, real, mask, rng, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(16):
            it.iteration(i, real, mask, rng, None)
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t1
        # the eight phases of the reference's profiler (Miscellaneous/train_time_profiler.py:186-314), on a second,
        # untimed-for-`value` period of 16 iterations with HIP events at the same boundaries; one stream (the teacher's
        # side stream off) so that a phase's time is its own kernels'
        from cagc.op import modconv as _mc
        overlap_saved, kd.OVERLAP_TEACHER = kd.OVERLAP_TEACHER, False
        side_saved, _mc._SIDE_LIMIT = _mc._SIDE_LIMIT, 0
        fork_saved, _mc.FORK_TORGB = _mc.FORK_TORGB, False
        it.phase_timer = kd.PhaseTimer()
        for i in range(16):
            it.iteration(i, real, mask, rng, None)
        phases = it.phase_timer.summary()
        it.phase_timer = None
        # the D step's MFMA entry points by themselves (one-stream, HIP events per launch): where train_D_d_forward / _backward go
        d_kernels = None
        try:
            zs_d = [torch.randn(bs, 512, device=dev)]
            it.d_step(real, zs_d)
            with KernelTimer(_lib) as ktd:
                for _ in range(2):
                    it.d_step(real, zs_d)
            aggd = ktd.summary()
            d_kernels = {k: {"ms_per_d_step": round(v[1] / 2, 3), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                             "what": ("weight gradient (direct convolution: K = pixels x batch)" if "wgrad" in k else
                                      "data gradient" if "dgrad" in k else "forward")}
                         for k, v in sorted(aggd.items(), key=lambda kv: -kv[1][1]) if v[2] > 0 and v[1] / 2 >= 0.3}
            d_kernels["_all_launches_ms_per_d_step"] = round(sum(v[1] for v in aggd.values()) / 2, 2)
        except Exception as e:  # noqa: BLE001 — a diagnostic
            print(f"[bench] D-step kernel pass failed ({type(e).__name__}: {e})", file=sys.stderr)
        kd.requires_grad(student, True)       # d_step froze the student: the later legs build generator steps on copies of it
        kd.requires_grad(disc, False)
        kd.OVERLAP_TEACHER = overlap_saved
        _mc._SIDE_LIMIT = side_saved
        _mc.FORK_TORGB = fork_saved
        full = {"value": round(16 * bs / dtf, 2), "unit": "images/s", "ms_per_iteration": round(dtf / 16 * 1e3, 2),
                "what": "D step + G/KD step + R1/16 + path-length/4 + EMA, bs16, eager launches (train.py:371-398 equivalent); "
                        "every convolution incl. the second-order passes and D's weight gradients on libcagc (no MIOpen)",
                "phases": phases, "d_step_mfma_entry_points": d_kernels,
                "phases_note": "GPU ms per call over 16 iterations (R1 runs in 1, the path-length regulariser in 4 of them); "
                               "train_G_d_forward = frozen-D forward + teacher forward + KD loss, as the reference brackets it"}
        del it, g_ema
        torch.cuda.empty_cache()
    proxy = None
    if world == 1 and not args.no_proxy and not os.environ.get("CAGC_BENCH_LOCAL_BS"):
        # Strong-scaling proxy (driver-visible): this one GPU at the per-GPU batch of an N-GPU run of the same global batch
        # 16, both launch modes, untimed-calibration style (3 warm-up + 8 steps each).  Upper bound on N-GPU efficiency
        # before communication: t(16) / (N * t(16/N)).
        proxy = {}
        for lb in (16, 8, 4, 2):
            mk = kd.ellipse_mask(lb, SIZE, dev)
            row = {}
            for m in ("graph", "eager"):
                try:
                    row[m + "_ms"] = round(_time_mode(kd, cd, student, teacher, disc, lb, mk, 1, dev, rng, gen, m), 3)
                except Exception as e:  # noqa: BLE001
                    row[m + "_ms"] = None
                    print(f"[bench] proxy bs {lb} {m} failed ({type(e).__name__}: {e})", file=sys.stderr)
                torch.cuda.empty_cache()
            best = min(v for v in row.values() if v is not None)
            row["n_gpus_equivalent"] = GLOBAL_BATCH // lb
            proxy[str(lb)] = row
        t16 = min(v for k, v in proxy["16"].items() if k.endswith("_ms") and v is not None)
        for lb in (8, 4, 2):
            tb = min(v for k, v in proxy[str(lb)].items() if k.endswith("_ms") and v is not None)
            proxy[str(lb)]["max_strong_scaling_efficiency"] = round(t16 / ((GLOBAL_BATCH // lb) * tb), 3)
    det = None
    if world == 1 and not args.no_proxy and not os.environ.get("CAGC_BENCH_LOCAL_BS"):
        # The same step in deterministic mode (cagc_set_tuning("deterministic", 1): no fp32-atomic K split, every backward reduction
        # through the order-independent sink — forward AND gradients bit-reproducible run to run): the price of reproducibility
        det = {}
        with _lib.tuning(deterministic=1):
            for lb in (16, 2):
                try:
                    det[str(lb)] = {"graph_ms": round(_time_mode(kd, cd, student, teacher, disc, lb, kd.ellipse_mask(lb, SIZE, dev), 1, dev, rng, gen, "graph"), 3)}
                except Exception as e:  # noqa: BLE001
                    det[str(lb)] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        for lb in (16, 2):
            base = (proxy or {}).get(str(lb), {}).get("graph_ms")
            if base and det[str(lb)].get("graph_ms"):
                det[str(lb)]["default_mode_graph_ms"] = base
                det[str(lb)]["overhead"] = round(det[str(lb)]["graph_ms"] / base - 1.0, 4)
    c3 = None
    if world == 1 and SIZE == 256 and not args.no_config3:
        # BASELINE configs[3] (1024 px pruned student [..,20,20,10,10] + full 1024 px teacher + D(1024), global batch 16 over 4
        # GPUs): ONE rank's share (per-GPU batch 4) on this GPU, HIP-graph replay, a few steps — the driver-visible figure
        try:
            s3, t3, d3 = kd.build_synthetic_workload(1024, dev, seed=0)
            m3 = kd.ellipse_mask(4, 1024, dev)
            st3 = kd.GraphedKDStep(s3, t3, d3, 4, m3, random_noise=True, world_size=1)
            for _ in range(2):
                st3.sample_and_step(4, m3, rng, gen)
            torch.cuda.synchronize()
            t3a = time.perf_counter()
            for _ in range(6):
                st3.sample_and_step(4, m3, rng, gen)
            torch.cuda.synchronize()
            dt3 = (time.perf_counter() - t3a) / 6
            c3 = {"value": round(4 / dt3, 2), "unit": "images/s", "ms_per_step": round(dt3 * 1e3, 3), "per_gpu_batch": 4, "steps": 6,
                  "what": "configs[3]: 1024px 70%-pruned student [154x10,77,77,39,39,20,20,10,10] + full 1024px teacher + D(1024) KD generator "
                          "step; one rank's share (per-GPU batch 4 of the 4-GPU global batch 16), HIP-graph replay; x4 = the job's upper bound"}
            del st3, s3, t3, d3
        except Exception as e:  # noqa: BLE001
            c3 = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    sweep = None
    if world == 1 and args.sweep:
        import gc
        gc.collect()
        torch.cuda.empty_cache()      # bs-64 fwd+bwd of the full generator: start from a released cache, whatever ran before
        # Rounds 3-4 called this leg "bimodal per process" (417 - 491 img/s against 522 - 554).  Round 5 (scripts/sweep_modes.py,
        # profiles/r05_sweep_episodes.log: ten consecutive 3-batch segments in each of 8 processes): it is not per process — about one
        # segment in eight runs 8 - 25 % slow and the next one is fast again, at the SAME in-kernel shader clock (2250 - 2290 MHz), and in a
        # slow episode exactly two entry points are slower, the two that are both MFMA- and HBM-heavy at batch 64: cagc_modconv_wgrad_demod
        # (66.5 -> 82.2 ms per batch) and the F(4x4) Winograd forward (40.0 -> 55.0 ms); every other kernel is within 2 %.  No allocation,
        # launch-plan or address difference goes with it (same process, same tensors).  Reading: a board-level power / memory-clock
        # management episode the shader-clock probe cannot see.  The leg therefore reports the MEDIAN batch as well as the mean.
        # BASELINE configs[4]: prune.py's content-aware saliency sweep over the FULL 256 px generator, bs 64 (forward +
        # backward incl. weight gradients of the 512-channel layers); bounded here to a few batches
        from cagc import prune
        requires = [p.requires_grad_(True) for p in teacher.parameters()]   # noqa: F841
        teacher.train()
        mfn = lambda im: kd.ellipse_mask(im.shape[0], SIZE, dev)
        nb = args.sweep
        prune.content_aware_scores(teacher, 64, 64, 0.05, mfn, dev)
        torch.cuda.synchronize()
        import ctypes as _ct2
        sw_clk = torch.zeros(2, device=dev)      # every 64th workgroup of the F(4x4) / register-direct launches adds its clock (two atomics): the
        if not args.no_sweep_clock:
            _lib.load().cagc_set_clock_probe(_ct2.c_void_p(sw_clk.data_ptr()))  # KD step's time is unchanged by it (30.09 vs 30.08 ms)
        t2 = time.perf_counter()
        per_batch = []
        for _ in range(nb):      # one bs-64 batch per call: per-batch wall times for the median (a transient slow episode hits 1 - 3 batches)
            tb = time.perf_counter()
            sc = prune.content_aware_scores(teacher, 64, 64, 0.05, mfn, dev)
            torch.cuda.synchronize()
            per_batch.append(time.perf_counter() - tb)
        dts = time.perf_counter() - t2
        _lib.load().cagc_set_clock_probe(None)
        med_b = sorted(per_batch)[len(per_batch) // 2]
        sweep = {"value": round(64 * nb / dts, 2), "value_median_batch": round(64 / med_b, 2), "slowest_batch_ms": round(max(per_batch) * 1e3, 1),
                 "median_batch_ms": round(med_b * 1e3, 1), "unit": "images/s", "batches": nb, "batch_size": 64,
                 "shader_clock_mhz": round(float(sw_clk[0] / sw_clk[1])) if float(sw_clk[1]) > 0 else None,
                 "what": "content-aware saliency sweep, full 256px generator fwd+bwd (271 GFLOP/img), on-device mask/noise/score",
                 "method": "one prune.content_aware_scores call per bs-64 batch, each followed by a device synchronise (per-batch wall times for the "
                           "median; since round 5 — rounds 1-4 timed ONE call over all batches, so their values are ~1 % higher for the same kernels)",
                 "direct_equivalent_tflops": round(64 * nb / dts * 271e9 / 1e12, 1),
                 "tflops_note": "direct_equivalent = 271 GFLOP/img (every conv as a direct convolution) x images/s: NOT a utilisation figure, it may exceed "
                                "the 157.3 TFLOP/s fp32-MFMA peak; executed = the MFMA flops the launches of one more (untimed) batch issue, over the median batch time",
                 "score_layers": len(sc)}
        try:      # executed flops of one batch (HIP-event-free: only the launch arguments are read)
            with KernelTimer(_lib) as kts:
                prune.content_aware_scores(teacher, 64, 64, 0.05, mfn, dev)
            sweep["executed_tflops"] = round(sum(r[3] for r in kts.records) / med_b / 1e12, 1)
        except Exception as e:  # noqa: BLE001
            sweep["executed_tflops"] = None
            print(f"[bench] sweep executed-flops pass failed ({type(e).__name__}: {e})", file=sys.stderr)
        teacher.eval()
        kd.requires_grad(teacher, False)
    cpu = None
    c0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            c0 = configs0_cpu_forward()
        except Exception as e:  # noqa: BLE001 — a secondary leg never takes the headline down
            c0 = {"error": f"{type(e).__name__}: {e}"}
        cpu = cpu_baseline(args.cpu_steps, args.cpu_batch)
        allc = os.cpu_count() or 1
        if allc > cpu["cores"] and args.cpu_all_cores:
            # SURVEY 8-d names torch.set_num_threads(os.cpu_count()); the grouped-conv CPU kernels oversubscribe badly beyond ~32
            # threads (round 3, 2x EPYC 9575F = 256 hardware threads: 0.0134 img/s), so it is timed only on request (+5 min)
            try:
                ac = cpu_baseline(1, 4, 60.0, threads=allc)
                cpu["all_cores"] = {"value": ac["value"], "unit": "images/s", "cores": allc, "sample": ac["sample"]}
            except Exception as e:  # noqa: BLE001
                cpu["all_cores"] = {"error": f"{type(e).__name__}: {e}", "cores": allc}

    if world == 1 and not args.no_roofline and mode == "graph" and roof is not None:
        # LAST leg on purpose: it captures one more HIP graph (with the clock probe set, so that the replays carry it), and a diagnostic must
        # not be able to influence a measured leg — so nothing runs after it.
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        try:
            roof["graph_replay_clock"] = _graph_replay_clock(kd, student, teacher, disc, bs, mask, dev, rng, gen)
        except Exception as e:  # noqa: BLE001 — a diagnostic, never a reason to lose the bench line
            print(f"[bench] graph-replay clock probe failed ({type(e).__name__}: {e})", file=sys.stderr)
            _lib.load().cagc_set_clock_probe(None)
            roof["graph_replay_clock"] = None

    if rank == 0:
        shape_txt = "[154x10,77,77,39,39]" if SIZE == 256 else "[154x10,77,77,39,39,20,20,10,10]"
        cfg_txt = "configs[1]" if SIZE == 256 else "configs[3]"
        out = {"metric": f"KD-retrain images/sec, {SIZE}px StyleGAN2 70%-pruned bs16", "value": round(value, 3),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "median_ms_per_step": None if median_ms is None else round(median_ms, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{SIZE}px StyleGAN2 70%-pruned student {shape_txt} + full teacher KD generator step, "
                                      f"global bs16 ({cfg_txt}); frozen D fwd+dgrad, masked-L1 KD, Adam; LPIPS/BiSeNet off"
                                      + (f"; {share}" if share else ""),
                          "global_batch": bs * world, "per_gpu_batch": bs, "parallelism": f"dp{world}", "launch_mode": mode, "launch_mode_calibration_ms": calib,
                          "student_params": n_params,
                          # what rank 0 saw of the job: the driver can check "RCCL saw N ranks" against its own launch
                          "world_size_env": world, "dist_initialized": dist.is_initialized(),
                          "dist_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "dist_backend": dist.get_backend() if dist.is_initialized() else None,
                          # graph mode: "graph" = bucketed RCCL all-reduces captured inside the one step graph, launched from the backward
                          # as each gradient bucket completes; "host" = one flat all-reduce between two graphs (fallback); eager: DDP buckets
                          "grad_collective": (getattr(step, "comm", None) if mode == "graph" else "ddp_buckets") if world > 1 else None,
                          "grad_collective_reason": getattr(step, "comm_reason", None) if (mode == "graph" and world > 1) else None,
                          "grad_buckets": len(getattr(step, "_buckets", [])) if (mode == "graph" and world > 1) else None,
                          "grad_bucket_bytes": bucket_bytes, "grad_bucket_allreduce_ms": bucket_ms,
                          "grad_allreduce_ms_sum": None if bucket_ms is None else round(sum(bucket_ms), 4),
                          "strict_comm": os.environ.get("CAGC_STRICT_COMM", "0") == "1"},
               "roofline": roof, "cpu_baseline": cpu, "full_iteration": full, "saliency_sweep": sweep,
               "strong_scaling_proxy_1gpu": proxy, "deterministic_mode": det, "config3_1024": c3, "configs0_cpu_forward": c0}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
