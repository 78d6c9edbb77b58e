"""The KD-retrain generator step — the measured unit (reference train.py:280-308 `G_Loss_BackProp`,
:145-184 `KD_loss` in 'Output_Only' mode, :203-206 non-saturating loss, :226-237 style mixing), plus the
synthetic workload of SURVEY.md §8-d.

One process per GPU; the student may be wrapped in DistributedDataParallel (cagc/distributed.py) — its gradient
all-reduce over RCCL is the path's only collective.  The content mask is supplied as a {0,1} tensor [B,1,H,W]
(the reference derives it from BiSeNet, whose weights are not available offline) and
LPIPS is off (`kd_lpips_lambda = 0`, weights unobtainable offline) — both stated in DESIGN.md."""
import math
import os
import random

import torch
from torch.nn import functional as F

from . import content_mask
from . import model as M
from . import prune
from .op import modconv as mc


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def g_nonsaturating_loss(fake_pred):
    return F.softplus(-fake_pred).mean()


def index_aware_mixing_noise(batch, latent_dim, prob, n_latent, device, rng=random, generator=None):
    """(list of 1 or 2 z, inject_index | None) — reference train.py:226-237."""
    if prob > 0 and rng.random() < prob:
        z = torch.randn(2, batch, latent_dim, device=device, generator=generator).unbind(0)
        return list(z), rng.randint(1, n_latent - 1)
    return [torch.randn(batch, latent_dim, device=device, generator=generator)], None


def ellipse_mask(batch, size, device, coverage=0.6):
    """Deterministic stand-in for the face-parsing mask: centred ellipse covering ~60 % of the pixels."""
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32),
                            indexing="ij")
    c = (size - 1) / 2
    a = size * 0.5 * math.sqrt(coverage * 4 / math.pi) * 1.1
    b = size * 0.5 * math.sqrt(coverage * 4 / math.pi) / 1.1
    m = (((yy - c) / a) ** 2 + ((xx - c) / b) ** 2 <= 1.0).float()
    return m.reshape(1, 1, size, size).repeat(batch, 1, 1, 1).to(device)


OVERLAP_TEACHER = os.environ.get("CAGC_OVERLAP_TEACHER", "1") == "1"


class KDStep:
    """student / teacher / discriminator + Adam, with `g_step` = one G_Loss_BackProp."""

    def __init__(self, student, teacher, discriminator, lr=0.002, g_reg_every=4, kd_l1_lambda=3.0, mixing=0.9,
                 latent=512, fused_adam=None, parsing_net=None, kd_mode="Output_Only", percept_loss=None,
                 kd_lpips_lambda=3.0, lpips_image_size=256):
        """parsing_net: callable image-batch -> 19-class logits [B,19,512,512] (or a tuple starting with them, BiSeNet's
        convention).  When given, the content mask is derived on device from the TEACHER's image every step
        (cagc.content_mask, reference train.py:155-158) and the `mask` argument of the step functions may be None.
        kd_mode: 'Output_Only' | 'Intermediate' (train.py:163-169).  percept_loss: optional callable (LPIPS in the
        reference, train.py:173-182; its weights are not obtainable offline, so it is a hook, off by default)."""
        assert kd_mode in ("Output_Only", "Intermediate")
        self.student, self.teacher, self.disc = student, teacher, discriminator
        # frozen use of D (generator step): bypass a DistributedDataParallel wrapper — its reducer expects a gradient for
        # every parameter of a forward it has seen, and D's parameters get none while frozen
        self.disc_frozen = discriminator.module if hasattr(discriminator, "module") else discriminator
        self.kd_l1_lambda, self.mixing, self.latent = kd_l1_lambda, mixing, latent
        self.parsing_net, self.kd_mode = parsing_net, kd_mode
        self.percept_loss, self.kd_lpips_lambda, self.lpips_image_size = percept_loss, kd_lpips_lambda, lpips_image_size
        self.teacher.eval()
        requires_grad(self.teacher, False)
        c = g_reg_every / (g_reg_every + 1)                      # lazy-regularisation correction, train.py:528-532
        params = [p for p in student.parameters()]
        dev = params[0].device
        kw = {}
        if fused_adam is None:
            fused_adam = dev.type == "cuda"
        if fused_adam:
            kw["fused"] = True
        self.optim = torch.optim.Adam(params, lr=lr * c, betas=(0.0 ** c, 0.99 ** c), **kw)
        self.n_latent = (student.module if hasattr(student, "module") else student).n_latent
        self.phase_timer = None

    def _mark(self, phase=None):
        if self.phase_timer is not None:
            self.phase_timer.mark(phase)

    def _forward_all(self, zs, inject_index, mask, student_noise=None, teacher_noise=None):
        """student, frozen D and teacher forward -> (D scores of the student image, student rgb list, teacher rgb list, mask)"""
        # The frozen teacher's forward is independent of the student / discriminator chain until the distillation loss:
        # on the GPU it runs on its own HIP stream, so its launches fill the CUs that the student's narrow (154/77/39
        # channel) and low-resolution layers leave idle, and kernel tails of one chain overlap the other.  Works eagerly
        # and under HIP-graph capture (fork / join become graph dependencies).
        overlap = zs[0].is_cuda and OVERLAP_TEACHER

        # only the 'Intermediate' distillation term reads the lower-resolution RGB outputs: without them the generators keep their
        # ToRGB chain private (its backward then runs beside the styled convs' backward, mc._ToRGB)
        want_list = self.kd_mode == "Intermediate"

        def run_teacher():
            with torch.no_grad():
                t_list = self.teacher(zs, return_rgb_list=want_list, inject_index=inject_index, noise=teacher_noise)
                if not want_list:
                    t_list = [t_list]
                m = mask
                if self.parsing_net is not None:     # on-device content mask from the teacher's image (train.py:155-158)
                    m = content_mask.teacher_content_mask(t_list[-1], self.parsing_net)
            return t_list, m

        if overlap:
            main = torch.cuda.current_stream()
            if getattr(self, "_teacher_stream", None) is None:
                self._teacher_stream = torch.cuda.Stream()
            side = self._teacher_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                teacher_list, mask = run_teacher()
        self._mark()
        fake_list = self.student(zs, return_rgb_list=want_list, inject_index=inject_index, noise=student_noise)
        if not want_list:
            fake_list = [fake_list]
        self._mark("train_G_g_forward")
        fake_pred = self.disc_frozen(fake_list[-1])
        if overlap:
            main.wait_stream(side)
            for t in teacher_list:
                t.record_stream(main)
            if torch.is_tensor(mask):
                mask.record_stream(main)
        else:
            teacher_list, mask = run_teacher()
        return fake_pred, fake_list, teacher_list, mask

    def _kd_term(self, fake_list, teacher_list, mask):
        fake_img, teacher_img = fake_list[-1], teacher_list[-1]
        if self.kd_mode == "Output_Only":
            # content-aware KD off (parsing_net None, no mask supplied; reference train.py:155,516-518): plain L1
            kd_l1 = self.kd_l1_lambda * (mc.masked_l1(fake_img, teacher_img, mask) if mask is not None
                                         else torch.mean(torch.abs(teacher_img - fake_img)))
        else:
            # 'Intermediate' (train.py:165-169): L1 over EVERY resolution's RGB output.  As in the reference the lists hold
            # the un-masked images — its masked copies only feed the Output_Only and LPIPS terms.
            kd_l1 = self.kd_l1_lambda * sum(torch.mean(torch.abs(t - s)) for t, s in zip(teacher_list, fake_list))
        if self.percept_loss is not None:
            # LPIPS hook (train.py:173-182): masked student vs masked teacher, pooled to `lpips_image_size` above it.  (In
            # 'Intermediate' mode the reference's loop variable leaves the teacher un-masked here; reproduced.)
            s_in = fake_img * mask if mask is not None else fake_img
            t_in = teacher_img if (self.kd_mode == "Intermediate" or mask is None) else teacher_img * mask
            if fake_img.shape[-1] > self.lpips_image_size:
                size = (self.lpips_image_size, self.lpips_image_size)
                s_in = F.interpolate(s_in, size=size, mode="bilinear", align_corners=False)
                t_in = F.interpolate(t_in, size=size, mode="bilinear", align_corners=False)
            self.last_kd_lpips = self.kd_lpips_lambda * torch.mean(self.percept_loss(s_in, t_in))
            kd_l1 = kd_l1 + self.last_kd_lpips          # second return value = the whole distillation term
        return kd_l1

    def g_losses(self, zs, inject_index, mask, student_noise=None, teacher_noise=None):
        """(g_loss, kd_l1, student image) as separate differentiable terms (reference train.py:296-304)."""
        fake_pred, fake_list, teacher_list, mask = self._forward_all(zs, inject_index, mask, student_noise, teacher_noise)
        return g_nonsaturating_loss(fake_pred), self._kd_term(fake_list, teacher_list, mask), fake_list[-1]

    def g_total(self, zs, inject_index, mask, student_noise=None, teacher_noise=None, grad_scale=1.0, unit_seed=False):
        """(total, g_loss, kd_l1): `total` = g_loss + kd_l1 is the tensor to differentiate (its gradients come out multiplied by
        grad_scale — the 1 / world_size of a data-parallel mean); the other two are detached values.
        On the GPU in the measured configuration (Output_Only, content mask, no LPIPS) the three losses and both backward seeds are ONE
        launch (cagc_gan_kd_loss_tail) instead of a chain of ~14 aten kernels; anything else composes the reference's terms.
        unit_seed=True (only `g_step` / `_fwd_bwd`, which own the `backward()` call) lets that launch write the seeds for the IMPLICIT
        unit upstream gradient and skips the multiply by it; the default honours whatever the caller backpropagates — a scaled
        loss (`(total / accum).backward()`, a GradScaler, `autograd.grad(grad_outputs=...)`) scales the gradients on both paths."""
        fake_pred, fake_list, teacher_list, mask = self._forward_all(zs, inject_index, mask, student_noise, teacher_noise)
        if (self.kd_mode == "Output_Only" and self.percept_loss is None and mask is not None
                and mc.gan_kd_loss_tail_ok(fake_pred, fake_list[-1], teacher_list[-1], mask)):
            return mc.gan_kd_loss_tail(fake_pred, fake_list[-1], teacher_list[-1], mask, self.kd_l1_lambda, grad_scale,
                                       unit_seed=unit_seed)
        g_loss, kd_l1 = g_nonsaturating_loss(fake_pred), self._kd_term(fake_list, teacher_list, mask)
        total = g_loss + kd_l1
        if grad_scale != 1.0:
            total = total * grad_scale + (total.detach() * (1.0 - grad_scale))      # value unchanged, gradients scaled
        return total, g_loss.detach(), kd_l1.detach()

    def g_step(self, zs, inject_index, mask, student_noise=None, teacher_noise=None):
        requires_grad(self.student, True)
        requires_grad(self.disc, False)
        total, g_loss, kd_l1 = self.g_total(zs, inject_index, mask, student_noise, teacher_noise, unit_seed=True)
        self._mark("train_G_d_forward")      # as in the reference's profiler: D forward + teacher forward + the KD loss
        self.optim.zero_grad(set_to_none=True)
        total.backward()
        self.optim.step()
        self._mark("train_G_g_backward")
        if total.is_cuda:
            mc.check_streamk_error(total.device)      # host-mapped word, no synchronisation (csrc/conv_streamk.h)
        if self.percept_loss is not None:
            lp = self.last_kd_lpips.detach()
            return {"g": g_loss, "kd_l1_loss": kd_l1 - lp, "kd_lpips_loss": lp}
        return {"g": g_loss, "kd_l1_loss": kd_l1}

    def sample_and_step(self, batch, mask, rng=random, generator=None):
        dev = mask.device
        zs, inj = index_aware_mixing_noise(batch, self.latent, self.mixing, self.n_latent, dev, rng, generator)
        return self.g_step(zs, inj, mask)


def d_logistic_loss(real_pred, fake_pred):
    """reference train.py:187-191."""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    """reference train.py:194-200 (double backward through D)."""
    (grad_real,) = torch.autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def accumulate(model_ema, model, decay=0.999):
    """EMA of the generator weights — reference train.py:124-129 (its add_(Number, Tensor) overload is gone from
    current PyTorch; same arithmetic)."""
    pe, pm = dict(model_ema.named_parameters()), dict(model.named_parameters())
    with torch.no_grad():
        torch._foreach_mul_(list(pe.values()), decay)
        torch._foreach_add_(list(pe.values()), [pm[k].detach() for k in pe], alpha=1 - decay)
    M.invalidate_caches(model_ema)


class PhaseTimer:
    """GPU time of the eight phases the reference's profiler reports (Miscellaneous/train_time_profiler.py:186-314):
    train_D_{g_forward, d_forward, d_backward}, reg_D, train_G_{g_forward, d_forward, g_backward}, reg_G — HIP events on the
    launch stream at the same boundaries (the reference brackets them with host clocks and no device sync).  Attach to a
    KDStep / TrainIteration as `.phase_timer`; `summary()` synchronises once."""

    def __init__(self):
        self.events = []        # (phase that ENDS here | None for a start mark, event)

    def mark(self, phase=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.events.append((phase, ev))

    def summary(self):
        torch.cuda.synchronize()
        tot, cnt = {}, {}
        for (_, e0), (ph, e1) in zip(self.events[:-1], self.events[1:]):
            if ph is not None:
                tot[ph] = tot.get(ph, 0.0) + e0.elapsed_time(e1)
                cnt[ph] = cnt.get(ph, 0) + 1
        return {k: {"calls": cnt[k], "ms_per_call": round(tot[k] / cnt[k], 3), "ms_total": round(tot[k], 2)} for k in tot}


class TrainIteration(KDStep):
    """The rest of one reference training iteration around the KD generator step (reference train.py:371-398):
    D step (:241-262), lazy R1 every `d_reg_every` (:264-278), G+KD step (KDStep.g_step), lazy path-length
    regulariser every `g_reg_every` (:310-338), EMA (:398).  Second-order passes (R1 through D, path length through
    G) run on the twice-differentiable composed ops (`composed_autograd`), whose upfirdn2d / fused-act members are
    still the HIP kernels."""

    def __init__(self, student, teacher, discriminator, g_ema=None, lr=0.002, r1=10.0, path_regularize=2.0,
                 path_batch_shrink=2, g_reg_every=4, d_reg_every=16, **kw):
        super().__init__(student, teacher, discriminator, lr=lr, g_reg_every=g_reg_every, **kw)
        self.g_ema = g_ema
        self.r1, self.path_regularize, self.path_batch_shrink = r1, path_regularize, path_batch_shrink
        self.g_reg_every, self.d_reg_every = g_reg_every, d_reg_every
        c = d_reg_every / (d_reg_every + 1)
        dparams = list(discriminator.parameters())
        kwd = {"fused": True} if dparams[0].device.type == "cuda" else {}
        self.d_optim = torch.optim.Adam(dparams, lr=lr * c, betas=(0.0 ** c, 0.99 ** c), **kwd)
        self.mean_path_length = 0.0
        self.accum = 0.5 ** (32 / (10 * 1000))

    def d_step(self, real_img, zs, inject_index=None, noise=None):
        requires_grad(self.student, False)
        requires_grad(self.disc, True)
        self._mark()
        with torch.no_grad():      # G is frozen here: nothing of its graph is needed (same values as the reference's :251)
            fake_img = (self.student.module if hasattr(self.student, "module") else self.student)(
                zs, inject_index=inject_index, noise=noise)
        self._mark("train_D_g_forward")
        fake_pred = self.disc(fake_img)
        real_pred = self.disc(real_img)
        d_loss = d_logistic_loss(real_pred, fake_pred)
        self._mark("train_D_d_forward")
        self.d_optim.zero_grad(set_to_none=True)
        d_loss.backward()
        self.d_optim.step()
        self._mark("train_D_d_backward")
        return {"d": d_loss.detach(), "real_score": real_pred.mean().detach(), "fake_score": fake_pred.mean().detach()}

    def d_reg(self, real_img):
        requires_grad(self.disc, True)
        self._mark()
        real_img = real_img.detach().requires_grad_(True)
        # double backward through D: its fused ops build a differentiable backward under create_graph=True (closed conv
        # family of op/conv_closure.py + the twice-differentiable upfirdn2d / fused-act ops) — all on libcagc
        real_pred = self.disc(real_img)
        r1_loss = d_r1_loss(real_pred, real_img)
        self.d_optim.zero_grad(set_to_none=True)
        (self.r1 / 2 * r1_loss * self.d_reg_every + 0 * real_pred[0]).backward()
        self.d_optim.step()
        self._mark("reg_D")
        return r1_loss.detach()

    def g_reg(self, zs, inject_index=None, noise=None):
        requires_grad(self.student, True)
        self._mark()
        fake_img, path_lengths = self.student(zs, PPL_regularize=True, inject_index=inject_index, noise=noise)
        path_mean = self.mean_path_length + 0.01 * (path_lengths.mean() - self.mean_path_length)
        path_loss = (path_lengths - path_mean).pow(2).mean()
        self.mean_path_length = path_mean.detach()
        self.optim.zero_grad(set_to_none=True)
        weighted = self.path_regularize * self.g_reg_every * path_loss
        if self.path_batch_shrink:
            weighted = weighted + 0 * fake_img[0, 0, 0, 0]
        weighted.backward()
        self.optim.step()
        self._mark("reg_G")
        return path_loss.detach(), path_lengths.detach()

    def ema(self):
        if self.g_ema is not None:
            accumulate(self.g_ema, self.student.module if hasattr(self.student, "module") else self.student, self.accum)

    def iteration(self, it, real_img, mask, rng=random, generator=None):
        """One full iteration on sampled latents (bench secondary figure)."""
        dev, B = mask.device, real_img.shape[0]
        out = {}
        zs, _ = (lambda z: (z, None))(  # D step uses plain mixing_noise (train.py:249): the forward draws the index
            list(torch.randn(2, B, self.latent, device=dev, generator=generator).unbind(0))
            if (self.mixing > 0 and rng.random() < self.mixing) else [torch.randn(B, self.latent, device=dev, generator=generator)])
        out.update(self.d_step(real_img, zs))
        if it % self.d_reg_every == 0:
            out["r1"] = self.d_reg(real_img)
        out.update(self.sample_and_step(B, mask, rng, generator))
        if it % self.g_reg_every == 0:
            pb = max(1, B // self.path_batch_shrink)
            zs = (list(torch.randn(2, pb, self.latent, device=dev, generator=generator).unbind(0))
                  if (self.mixing > 0 and rng.random() < self.mixing) else [torch.randn(pb, self.latent, device=dev, generator=generator)])
            out["path"], _ = self.g_reg(zs)
        self.ema()
        return out


class _CaptureFailed(RuntimeError):
    """The HIP-graph capture of a step failed (its __cause__ says why); raised only from inside the capture proper."""


class GraphedKDStep(KDStep):
    """The same step replayed from a HIP graph (torch.cuda.CUDAGraph): latent sampling, student / teacher / D forward, the loss
    tail, the whole backward, the gradient all-reduce and Adam.  This removes the ~1400 per-step host launches (the step is
    launch-bound at small per-GPU batch) — the 'HIP streams and graphs instead of a tracing compiler' of the design brief.  Style
    mixing uses a device-side index so shapes are static (Generator._synthesize).

    Data parallel (reference train.py:522-525 DistributedDataParallel, Miscellaneous/distributed.py:57-66): gradients live in ONE flat
    buffer laid out in the order in which backward produces them (probed once, eagerly) and cut into `n_buckets` contiguous buckets;
    the post-accumulate hook of whichever parameter completes a bucket gathers the bucket into its slice and issues its RCCL all-reduce FROM INSIDE
    the captured backward (`comm='graph'`: the collective is a graph node on RCCL's stream that depends only on that bucket — it runs
    beside the rest of the backward, and Adam waits for all of them; the 1 / world_size of the mean is folded into the backward seed
    by the loss tail).  If capturing a collective fails on some rank, every rank falls back to `comm='host'`: two graphs with one
    flat all-reduce between them (the round-1..4 form).  22.3 MB of gradients at 256 px."""

    def __init__(self, student, teacher, discriminator, batch, mask, random_noise=True, world_size=1, always_reduce=False,
                 comm="auto", n_buckets=4, **kw):
        """always_reduce: issue the collectives even at world size 1 (RCCL smoke test on a 1-GPU box: the same graph nodes / stream
        ordering as a multi-GPU run).  comm: 'auto' (graph, else host), 'graph', 'host'."""
        kw.setdefault("fused_adam", True)
        super().__init__(student, teacher, discriminator, **kw)
        assert comm in ("auto", "graph", "host")
        self.always_reduce = always_reduce
        for g in self.optim.param_groups:
            g["capturable"] = True
        if len(self.optim.param_groups) != 1 or self.optim.param_groups[0].get("amsgrad") or self.optim.param_groups[0].get("maximize"):
            raise ValueError("GraphedKDStep captures ONE flat Adam: a single param group without amsgrad / maximize")
        self.batch, self.world = batch, world_size
        self._reduce = world_size > 1 or always_reduce
        self._grad_scale = 1.0 / world_size
        dev = mask.device
        self.mask = mask.clone()
        self.z = torch.zeros(2, batch, self.latent, device=dev)
        self.inj = torch.full((1,), self.n_latent, device=dev, dtype=torch.long)
        self._inj_host = torch.zeros(1, dtype=torch.long).pin_memory()
        self.random_noise = random_noise
        base = student.module if hasattr(student, "module") else student
        if not random_noise:
            shp = lambda i: (batch, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2))
            self.s_noise = [torch.zeros(shp(i), device=dev) for i in range(base.num_layers)]
            self.t_noise = [torch.zeros(shp(i), device=dev) for i in range(base.num_layers)]
        else:
            self.s_noise = self.t_noise = None
        self._in_graph_comm = False
        self._works = []
        self._params = self._probe_grad_order([p for p in self.student.parameters()])
        # every parameter starts at a 16-byte boundary of the flat buffers (the kernels' vector loads want aligned weights: p.data is
        # re-pointed into the flat parameter); the <= 3 padding elements behind a parameter hold zeros (zero gradient, zero weight)
        self._offs, off = [], 0
        for p in self._params:
            self._offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        n_flat = off
        self.flat_grad = torch.zeros(n_flat, device=dev)
        self._pad = torch.zeros(4, device=dev)
        self._grad_views = [self.flat_grad[o:o + p.numel()].view_as(p) for o, p in zip(self._offs, self._params)]
        self._plan_buckets(max(1, int(n_buckets)))
        self._flatten_optimizer(n_flat, dev)
        self.losses = None
        self.comm = None
        if comm == "auto" and os.environ.get("CAGC_GRAPH_COMM") in ("graph", "host"):
            comm = os.environ["CAGC_GRAPH_COMM"]      # operator override: "host" = never capture a collective
        modes = ["graph", "host"] if comm == "auto" else [comm]
        if not self._reduce:
            modes = ["graph"]                 # nothing to communicate: one graph
        elif comm == "auto":
            import torch.distributed as dist
            if dist.get_backend() != "nccl":  # only RCCL collectives are stream-ordered device work that a HIP graph can hold
                modes = ["host"]
        # The eager warm-up (allocator pools, lazy initialisation, Adam state) runs ONCE, outside any fallback: an out-of-memory or a
        # kernel error there is a real error and propagates.  Only a failure of the stream capture itself selects the next mode, and
        # CAGC_STRICT_COMM=1 turns even that into an error (the first multi-GPU run should fail loudly rather than change mode).
        strict = os.environ.get("CAGC_STRICT_COMM", "0") == "1"
        self.comm_reason = ("no collective (world size 1)" if not self._reduce else
                            f"requested comm={comm!r}" if len(modes) == 1 and comm != "auto" else
                            "backend is not nccl/RCCL: collectives cannot be captured" if modes == ["host"] else
                            "collectives captured inside the step graph")
        self._warm_up()
        err = None
        for m in modes:
            ok = 1.0
            try:
                self._capture(m)
            except _CaptureFailed as e:      # the collective (or anything else) could not be captured: a supported outcome
                if m == modes[-1] or strict:
                    raise RuntimeError(f"GraphedKDStep: capturing the step with comm={m!r} failed"
                                       + (" (CAGC_STRICT_COMM=1: no fallback)" if strict and m != modes[-1] else "")) from e.__cause__
                err, ok = e.__cause__, 0.0
            if world_size > 1:                # every rank must take the same branch: the modes issue different collectives
                import torch.distributed as dist
                okt = torch.tensor([ok], device=dev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                ok = float(okt.item())
            if ok > 0:
                self.comm = m
                break
            why = f"{type(err).__name__}: {err}" if err else "failed on another rank"
            if strict:
                raise RuntimeError(f"GraphedKDStep: capturing the step with comm={m!r} failed on some rank ({why}); CAGC_STRICT_COMM=1: no fallback")
            self.comm_reason = f"fallback: capturing the collectives inside the graph failed ({why})"
            import sys
            print(f"[cagc] GraphedKDStep: capturing the collectives inside the graph failed ({why}); "
                  "falling back to one flat all-reduce between two graphs", file=sys.stderr)
        assert self.comm is not None

    def close(self):
        """Detach this step from the student: remove the bucket hooks (they hold `self`, i.e. the graphs and the flat buffers).
        Call before building another GraphedKDStep on the same student (the 'construct after loading and re-capture' resume path)."""
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks = []
        self._armed = False

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 — interpreter shutdown
            pass

    # ---- gradient layout ------------------------------------------------------------------------------------------------------
    def _probe_grad_order(self, params):
        """One eager forward / backward with post-accumulate hooks: the order in which backward finishes the parameters' gradients
        (parameters that receive none come last).  Leaves weights, gradients and optimiser untouched."""
        order, hooks = [], []
        index = {id(p): i for i, p in enumerate(params)}
        requires_grad(self.student, True)      # (a preceding D step leaves the student frozen: hooks want leaves that require grad)
        requires_grad(self.disc, False)
        for p in params:
            hooks.append(p.register_post_accumulate_grad_hook(lambda q: order.append(index[id(q)])))
        try:
            z = [torch.randn_like(self.z[0]), torch.randn_like(self.z[1])]
            inj = torch.full_like(self.inj, max(1, self.n_latent // 2))      # with style mixing: both passes through the mapping network
            total, _, _ = self.g_total(z, inj, self.mask, self.s_noise, self.t_noise, unit_seed=True)
            total.backward()
        finally:
            for h in hooks:
                h.remove()
            for p in params:
                p.grad = None
        seen = set(order)
        order += [i for i in range(len(params)) if i not in seen]
        return [params[i] for i in order]

    def _plan_buckets(self, n_buckets):
        """contiguous slices of the flat buffer of ~equal size, cut at parameter boundaries: (lo, hi, first param, last param + 1)"""
        sizes = [(p.numel() + 3) // 4 * 4 for p in self._params]      # padded extents (16-byte aligned starts)
        total, target = sum(sizes), sum(sizes) / n_buckets
        self._buckets, lo, first, acc = [], 0, 0, 0
        for i, n in enumerate(sizes):
            acc += n
            if (acc - lo >= target and len(self._buckets) < n_buckets - 1) or i == len(sizes) - 1:
                self._buckets.append((lo, acc, first, i + 1))
                lo, first = acc, i + 1
        assert self._buckets[-1][1] == total
        # a bucket closes when ALL of its parameters have their gradient — counted, not "when its last parameter in the probed order
        # fires": the order inside a bucket changes with the autograd graph (style mixing adds a second pass through the mapping network)
        self._bucket_of = {}
        for k, bk in enumerate(self._buckets):
            for q in self._params[bk[2]:bk[3]]:
                self._bucket_of[id(q)] = k
        # The hooks hold only a weak reference to this step, and a step that is re-built on the same student first removes its
        # predecessor's hooks: an abandoned step (and its graphs / flat buffers) is then free to be collected, and an eager backward
        # pays one Python call per parameter, not one per step ever constructed.
        import weakref
        requires_grad(self.student, True)
        base = self.student.module if hasattr(self.student, "module") else self.student
        prev = base.__dict__.get("_cagc_graphed_step")
        prev = prev() if prev is not None else None
        if prev is not None and prev is not self:
            prev.close()
        base.__dict__["_cagc_graphed_step"] = weakref.ref(self)
        me = weakref.ref(self)

        def hook(p):
            step = me()
            if step is not None:
                step._on_grad(p)
        self._hooks = [p.register_post_accumulate_grad_hook(hook) for p in self._params]
        self._armed = False

    def _on_grad(self, p):
        if self._armed:
            k = self._bucket_of[id(p)]
            self._arrived[k] += 1
            if self._arrived[k] == self._buckets[k][3] - self._buckets[k][2]:
                self._close_bucket(k)

    def _close_bucket(self, k):
        """gather bucket k's fresh gradients into its slice of the flat buffer and, under comm='graph', launch its all-reduce"""
        lo, hi, first, last = self._buckets[k]
        grads = []
        for q in self._params[first:last]:
            grads.append((q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1))
            if q.numel() % 4:
                grads.append(self._pad[:4 - q.numel() % 4])
        torch.cat(grads, out=self.flat_grad[lo:hi])
        self._closed[k] = True
        if self._in_graph_comm:
            import torch.distributed as dist
            self._works.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def _flatten_optimizer(self, n_flat, dev):
        """ONE Adam launch instead of four: the student's parameters become views of one flat buffer (`p.data` re-pointed, values
        kept), and the captured optimiser is `torch.optim.Adam` (fused, capturable — the same elementwise arithmetic) over that
        single flat parameter with the flat gradient: multi_tensor_apply over 135 small tensors took 0.21 ms per step at every batch
        size, the flat one ~0.04.  `self.optim` — the per-parameter optimiser every caller knows — stays as a VIEW for reading:
        its moment tensors are slices of the flat moments and its `step` entries all alias the one captured counter.  That aliasing
        must never reach a checkpoint (a plain Adam loaded from it would advance the shared counter once per parameter):
        `optim_state_dict()` exports per-parameter clones, and `checkpoint.save_checkpoint` uses it."""
        with torch.no_grad():
            flat_p = torch.zeros(n_flat, device=dev)
            for o, p in zip(self._offs, self._params):
                flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat_p[o:o + p.numel()].view_as(p)
        M.invalidate_caches(self.student)
        self._flat_param = torch.nn.Parameter(flat_p)
        self._flat_param.grad = self.flat_grad
        g0 = self.optim.param_groups[0]
        self._flat_optim = torch.optim.Adam([self._flat_param], lr=g0["lr"], betas=tuple(g0["betas"]), eps=g0["eps"],
                                            weight_decay=g0["weight_decay"], fused=True, capturable=True)
        flat_m, flat_v = torch.zeros(n_flat, device=dev), torch.zeros(n_flat, device=dev)
        step = torch.zeros((), dtype=torch.float32, device=dev)
        self._flat_optim.state[self._flat_param] = {"step": step, "exp_avg": flat_m, "exp_avg_sq": flat_v}
        for off, p in zip(self._offs, self._params):
            n = p.numel()
            self.optim.state[p] = {"step": step, "exp_avg": flat_m[off:off + n].view_as(p), "exp_avg_sq": flat_v[off:off + n].view_as(p)}

    def optim_state_dict(self):
        """`optim.state_dict()` with every tensor CLONED per parameter (own `step` counters, own moments): what a checkpoint may hold.
        Loadable by a plain torch.optim.Adam (eager KDStep / TrainIteration, the reference's train.py) and by `load_optim_state`."""
        sd = self.optim.state_dict()
        sd["state"] = {k: {n: (v.detach().clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()}
        return sd

    def _fwd_bwd(self):
        if self.random_noise:
            self.z.normal_()
        total, g_loss, kd_l1 = self.g_total([self.z[0], self.z[1]], self.inj, self.mask, self.s_noise, self.t_noise,
                                            grad_scale=self._grad_scale, unit_seed=True)
        # Gradients are produced into fresh tensors (grad = None: autograd assigns instead of launching one accumulate-add per
        # parameter into a pre-zeroed buffer); each bucket is gathered into the flat all-reduce / Adam buffer by ONE concatenation as
        # soon as its last gradient exists (_on_grad).
        for p in self._params:
            p.grad = None
        self._closed = [False] * len(self._buckets)
        self._arrived = [0] * len(self._buckets)
        self._works = []
        self._armed = True
        try:
            total.backward()
        finally:
            self._armed = False
        for k, done in enumerate(self._closed):      # buckets with a parameter that received no gradient
            if not done:
                self._close_bucket(k)
        for w in self._works:                        # Adam (and everything after) waits for the collectives: the join of RCCL's stream
            w.wait()
        self._works = []
        for p, v in zip(self._params, self._grad_views):
            p.grad = v
        return torch.stack([g_loss, kd_l1])

    def _warm_up(self):
        requires_grad(self.student, True)
        requires_grad(self.disc, False)
        # The warm-up (allocator pools, lazy inits, Adam state creation) takes real optimiser steps: snapshot the
        # student and restore it afterwards, so that capture leaves the weights / Adam moments / step counters exactly
        # as it found them (and replicas that started identical stay identical — no collective runs in the warm-up).
        params = self._params
        snap = [p.detach().clone() for p in params]
        self._in_graph_comm = False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._fwd_bwd()
                self._flat_optim.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, s in zip(params, snap):
                p.copy_(s)
            st = self._flat_optim.state[self._flat_param]
            for v in st.values():     # in place: the captured Adam graph keeps these tensors
                if torch.is_tensor(v):
                    v.zero_()
            self.flat_grad.zero_()
        if self.world > 1:                            # belt and braces: every replica starts from rank 0's weights
            import torch.distributed as dist
            dist.broadcast(self._flat_param.data, src=0)

    def _capture(self, mode):
        """Capture the step under `mode`; a failure INSIDE the stream capture raises _CaptureFailed (the caller may fall back)."""
        requires_grad(self.student, True)
        requires_grad(self.disc, False)
        self.graph_fb = self.graph_opt = None
        if mode == "graph":       # ONE graph: forward, backward with the bucket collectives inside it, Adam
            self._in_graph_comm = self._reduce
            g = torch.cuda.CUDAGraph()
            try:
                # thread-local capture-error mode while collectives are captured: RCCL's watchdog thread may query events of earlier
                # (already finished) collectives while this thread captures; under the default "global" mode such a query from ANOTHER
                # thread invalidates the capture.  Kernels launched by the autograd worker thread on the capturing stream are captured
                # in either mode (capture is a property of the stream).
                with torch.cuda.graph(g, capture_error_mode="thread_local" if self._in_graph_comm else "global"):
                    self.losses = self._fwd_bwd()
                    self._flat_optim.step()
            except Exception as e:      # noqa: BLE001 — whatever invalidated the capture (RCCL refusing to be captured, ...)
                self._in_graph_comm = False
                torch.cuda.synchronize()
                raise _CaptureFailed(mode) from e
            self.graph_fb = g
        else:                     # two graphs around one host-issued flat all-reduce
            self._in_graph_comm = False
            try:
                self.graph_fb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_fb):
                    self.losses = self._fwd_bwd()
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt):
                    self._flat_optim.step()
            except Exception as e:      # noqa: BLE001
                torch.cuda.synchronize()
                raise _CaptureFailed(mode) from e

    def replay(self, inject_index=None):
        """One step on whatever the static buffers hold.  inject_index: int in 1..n_latent-1, or None = no mixing."""
        mc.check_streamk_error(self.inj.device)      # what earlier replays left in the host-mapped error word (no synchronisation)
        self._inj_host[0] = self.n_latent if inject_index is None else int(inject_index)
        self.inj.copy_(self._inj_host, non_blocking=True)
        self.graph_fb.replay()
        if self.graph_opt is not None:
            if self._reduce:
                import torch.distributed as dist
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)      # (the mean's 1 / world is in the backward seed)
            self.graph_opt.replay()
        # the replayed Adam rewrites the weights without bumping Tensor._version: frozen uses of this student between
        # replays (TrainIteration.d_step, an eval after requires_grad_(False)) must not see stale packed weights
        M.invalidate_caches(self.student)
        return {"g": self.losses[0], "kd_l1_loss": self.losses[1]}

    def load_optim_state(self, state_dict):
        """Resume: copy a saved Adam state (`optim.state_dict()`, e.g. checkpoint['g_optim']) INTO the tensors the captured
        optimiser graph holds.  `optim.load_state_dict` after capture would swap those tensors for new ones and the graph
        would keep updating the old ones — unsupported; use this (or construct after loading and re-capture)."""
        saved = state_dict["state"]
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        params = [p for g in self.optim.param_groups for p in g["params"]]
        assert len(ids) == len(params), "optimizer state does not match this student's parameters"
        # hyper-parameters are baked into the captured Adam graph: a checkpoint written with others cannot be honoured here
        for gs, gl in zip(state_dict["param_groups"], self._flat_optim.param_groups):      # what the graph actually captured
            for key in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
                if key not in gs:
                    continue
                a, b = gs[key], gl[key]
                same = (all(abs(float(x) - float(y)) <= 1e-12 * max(1.0, abs(float(y))) for x, y in zip(a, b))
                        if isinstance(a, (tuple, list)) else (bool(a) == bool(b) if isinstance(b, bool) else abs(float(a) - float(b)) <= 1e-12 * max(1.0, abs(float(b)))))
                if not same:
                    raise ValueError(f"GraphedKDStep.load_optim_state: saved Adam {key} = {a} differs from the captured graph's {b}; "
                                     "construct the step with the checkpoint's hyper-parameters (they are captured, not reloadable)")
        with torch.no_grad():
            for i, p in zip(ids, params):
                if i not in saved:
                    continue
                live = self.optim.state[p]
                for k, v in saved[i].items():
                    if torch.is_tensor(v):
                        assert k in live and live[k].shape == v.shape, f"optimizer state '{k}' mismatch"
                        live[k].copy_(v.to(live[k].device, live[k].dtype))
                    elif isinstance(v, (int, float)) and not isinstance(v, bool):
                        # torch 1.6 (the reference's pin, README.md:58-61) stores Adam's `step` as a Python int: the captured
                        # capturable-Adam graph reads it from a device tensor — without this the bias correction restarts at t = 1
                        assert k in live and torch.is_tensor(live[k]), f"optimizer state '{k}' has no live tensor"
                        live[k].fill_(float(v))
        M.invalidate_caches(self.student)

    def sample_and_step(self, batch=None, mask=None, rng=random, generator=None):
        mix = self.mixing > 0 and rng.random() < self.mixing
        return self.replay(rng.randint(1, self.n_latent - 1) if mix else None)

    def g_step(self, zs, inject_index, mask, student_noise=None, teacher_noise=None):
        """Explicit-input form (parity tests): copy into the static buffers, replay."""
        assert not self.random_noise and student_noise is not None and teacher_noise is not None
        self.z[0].copy_(zs[0])
        self.z[1].copy_(zs[1] if len(zs) > 1 else zs[0])
        self.mask.copy_(mask)
        for dst, src in zip(self.s_noise + self.t_noise, list(student_noise) + list(teacher_noise)):
            dst.copy_(src.expand_as(dst))
        return self.replay(inject_index if len(zs) > 1 else None)


def build_synthetic_workload(size=256, device="cpu", seed=0, remove_ratio=0.7, style_dim=512, n_mlp=8,
                             noise_weight=0.1):
    """Teacher = seeded random-init full Generator with every noise weight set to 0.1 (init is 0 and would hide
    the noise path); student = teacher sliced to the uniform 70 %-pruned shape with masks from seeded random
    scores; D = seeded random-init Discriminator.  (SURVEY.md §8-d "Synthetic inputs".)"""
    import numpy as np
    torch.manual_seed(seed)
    teacher = M.Generator(size, style_dim, n_mlp)
    with torch.no_grad():
        for n, p in teacher.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(noise_weight)
    sd = teacher.state_dict()
    shape = prune.network_shape(sd)
    rng = np.random.RandomState(seed + 1)
    masks = prune.masks_from_scores([rng.rand(c) for c in shape], shape, prune.uniform_remove_list(shape, remove_ratio))
    ssd = prune.mask_generator_state_dict(sd, masks)
    student = M.Generator(size, style_dim, n_mlp, generator_net_shape=prune.network_shape(ssd))
    student.load_state_dict(ssd, strict=True)
    disc = M.Discriminator(size)
    return student.to(device), teacher.to(device), disc.to(device)
