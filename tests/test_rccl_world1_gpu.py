"""-m gpu: RCCL smoke on the hardware there is.  A 1-GPU box cannot run a multi-GPU collective, but it can
prove that `backend="nccl"` (= librccl on ROCm) initialises, that the student-gradient all-reduce runs both ways `GraphedKDStep`
issues it — `comm="graph"`: four bucket collectives CAPTURED INSIDE the one forward / backward / Adam graph, each launched by the
post-accumulate hook of its bucket's last gradient (no host-side collective between replays), and the fall-back
`comm="host"`: graph_fb -> all_reduce(flat_grad) -> graph_opt — and that
DistributedDataParallel's bucket hooks fire from the custom autograd nodes while the teacher runs on its side stream — at world
size 1 the reduced gradient must equal the local one, so the results are checked against the same steps without a process
group.  Replaces the reference's nn.DataParallel (train.py:522-525; intent of Miscellaneous/distributed.py:44-66)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(port):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd"), os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import cagc.model as M
    from cagc import _lib
    from cagc import distributed as cd
    from cagc import kd
    from oracle import ref_model
    from _util import load_json, load_npz, sub

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    g, meta = load_npz("kd_step_tiny"), load_json("kd_step_tiny_meta")

    def build():
        student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
        student.load_state_dict(sub(g, "student_sd/"), strict=True)
        teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
        teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
        disc = M.Discriminator(32)
        disc.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
        return student.to(dev), teacher.to(dev), disc.to(dev)

    cu = lambda t: t.to(dev)
    B = g["mask"].shape[0]

    def inputs(st, n):
        p = f"step{st['step']}/"
        return ([cu(g[p + f"z{i}"]) for i in range(st["n_z"])], st["inject_index"], cu(g["mask"]),
                [cu(g[p + f"student_noise{i}"]) for i in range(n)], [cu(g[p + f"teacher_noise{i}"]) for i in range(n)])

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))

    # a plain collective first: librccl loads, a communicator exists, the result is right
    t = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t[12345]) == 12345.0

    with _lib.tuning(deterministic=1):     # bit-reproducible steps: with / without the collective must agree exactly
        # (1) HIP-graph step with the RCCL all-reduces captured inside the graph (comm="graph") and, the fall-back form, as one host-issued
        #     collective between the forward/backward graph and the Adam graph (comm="host"): both BIT-EQUAL to the collective-free step
        s0, t0, d0 = build()
        ref = kd.GraphedKDStep(s0, t0, d0, B, cu(g["mask"]), random_noise=False, latent=24)
        assert ref.comm == "graph" and ref.graph_opt is None, "no collective: one graph"
        worst = 0.0
        for comm in ("graph", "host"):
            s1, t1, d1 = build()
            red = kd.GraphedKDStep(s1, t1, d1, B, cu(g["mask"]), random_noise=False, latent=24, world_size=1, always_reduce=True, comm=comm)
            assert red.comm == comm and len(red._buckets) == 4 and (red.graph_opt is None) == (comm == "graph")
            sr, tr, dr = build()
            again = kd.GraphedKDStep(sr, tr, dr, B, cu(g["mask"]), random_noise=False, latent=24)
            for st in meta["steps"]:
                la = again.g_step(*inputs(st, sr.num_layers))
                lb = red.g_step(*inputs(st, s1.num_layers))
            torch.cuda.synchronize()
            for (n, a), (_, b) in zip(sr.named_parameters(), s1.named_parameters()):
                assert torch.equal(a.detach(), b.detach()), f"comm={comm}: parameter {n} differs from the collective-free replay ({rel(b.detach(), a.detach()):.2e})"
            assert float(la["g"]) == float(lb["g"]) and float(la["kd_l1_loss"]) == float(lb["kd_l1_loss"])
        # (2) eager step under DistributedDataParallel: bucket hooks from the custom autograd nodes, teacher on its side stream
        s2, t2, d2 = build()
        s3, t3, d3 = build()
        plain = kd.KDStep(s2, t2, d2, latent=24)
        ddp = cd.wrap_student(s3, dev, force=True)
        assert type(ddp).__name__ == "DistributedDataParallel"
        wrapped = kd.KDStep(ddp, t3, d3, latent=24)
        for st in meta["steps"]:
            plain.g_step(*inputs(st, s2.num_layers))
            wrapped.g_step(*inputs(st, s3.num_layers))
        torch.cuda.synchronize()
        worst2 = max(rel(b.detach(), a.detach()) for (_, a), (_, b) in zip(s2.named_parameters(), s3.named_parameters()))
        assert worst2 <= 1e-6, f"DDP (world 1, RCCL) step differs from the unwrapped step: {worst2:.2e}"
        # (3) the fallback ladder: a failure INSIDE the capture of comm="graph" selects comm="host" and records why; under
        #     CAGC_STRICT_COMM=1 it is an error instead; an error in the eager WARM-UP (not a capture problem) always propagates
        real_capture = kd.GraphedKDStep._capture

        def failing_capture(self, mode):
            if mode == "graph":
                raise kd._CaptureFailed(mode) from RuntimeError("injected: RCCL refused to be captured")
            return real_capture(self, mode)

        kd.GraphedKDStep._capture = failing_capture
        try:
            s4, t4, d4 = build()
            fb = kd.GraphedKDStep(s4, t4, d4, B, cu(g["mask"]), random_noise=False, latent=24, world_size=1, always_reduce=True, comm="auto")
            assert fb.comm == "host" and fb.graph_opt is not None and "fallback" in fb.comm_reason and "injected" in fb.comm_reason, fb.comm_reason
            for st in meta["steps"]:
                lf = fb.g_step(*inputs(st, s4.num_layers))
            torch.cuda.synchronize()
            assert float(lf["g"]) == float(lb["g"])           # the fallback computes the same step
            os.environ["CAGC_STRICT_COMM"] = "1"
            try:
                s5, t5, d5 = build()
                try:
                    kd.GraphedKDStep(s5, t5, d5, B, cu(g["mask"]), random_noise=False, latent=24, world_size=1, always_reduce=True, comm="auto")
                    raise AssertionError("CAGC_STRICT_COMM=1 did not turn the capture failure into an error")
                except RuntimeError as e:
                    assert "CAGC_STRICT_COMM" in str(e), str(e)
            finally:
                del os.environ["CAGC_STRICT_COMM"]
        finally:
            kd.GraphedKDStep._capture = real_capture
        real_warm = kd.GraphedKDStep._warm_up

        def failing_warm(self):
            raise RuntimeError("injected: out of memory in the warm-up")

        kd.GraphedKDStep._warm_up = failing_warm
        try:
            s6, t6, d6 = build()
            try:
                kd.GraphedKDStep(s6, t6, d6, B, cu(g["mask"]), random_noise=False, latent=24, world_size=1, always_reduce=True, comm="auto")
                raise AssertionError("a warm-up error was swallowed by the comm fallback")
            except RuntimeError as e:
                assert "warm-up" in str(e)
        finally:
            kd.GraphedKDStep._warm_up = real_warm
    dist.destroy_process_group()
    print(f"RCCL_WORLD1_OK graph bit-equal (in-graph and host collectives) ddp {worst2:.1e}; fallback ladder ok")


def test_rccl_world1_graph_allreduce_and_ddp_hooks():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29800 + (os.getpid() % 150)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", str(port)], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    print(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "worker":
    _worker(int(sys.argv[2]))
