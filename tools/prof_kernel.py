#!/usr/bin/env python3
"""Run only the native hot-path call (prep -> main -> finalize) for rocprofv3 counter collection.

    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_kernel.py --workload C2 --iters 20
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU ... -- python3 tools/prof_kernel.py
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--num-pix", type=int, default=None)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--mode", default="grad", choices=["grad", "fwd", "img", "simpair", "lstsq"])
    ap.add_argument("--direct", action="store_true", help="C3: direct (non-table) shapelets")
    args = ap.parse_args()
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    kw = dict(num_pix=args.num_pix, batch=args.batch)
    if args.workload.upper() == "C3":
        kw["interpolate"] = not args.direct
    wl = workloads.make(args.workload, **kw)
    if args.mode == "lstsq":  # linear-amplitude solve: end-to-end time of gl_lstsq_fwd (coefficients)
        c2 = workloads.make("C2", num_pix=wl.sim_config.num_pix, batch=1)
        obs, _, _ = workloads.synthetic_observation(c2, LensSimulator)
        err = torch.sqrt(wl.background_rms ** 2 + obs.clamp_min(0) / wl.exp_time).contiguous()
        sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
        packed = sim.pack(wl.prior.sample(wl.batch, seed=0)).contiguous()
        for _ in range(2):
            sim._model.lstsq(packed, obs, err, 7, want="coeffs")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            sim._model.lstsq(packed, obs, err, 7, want="coeffs")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print(f"{wl.name} B={wl.batch} N={sim._model.N} D={sim._model.num_linear()} lstsq: {ms:.3f} ms per solve "
              f"-> {wl.batch / (ms * 1e-3):.0f} solves/s")
        return
    obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = sim.pack(wl.prior.sample(wl.batch, seed=0)).contiguous()
    m = sim._model
    if args.mode == "simpair":
        # the simulate() boundary itself: gl_simulate_fwd writes the [B,H,W] image, gl_simulate_bwd reads its cotangent --
        # the pair that really moves the B1 bytes (8N + 8P per sample) through HBM
        gimg = torch.randn((wl.batch, m.out_h, m.out_w), device=packed.device)
        m.set_timing(2 * args.iters)
        for i in range(args.iters):
            m.simulate_fwd(packed)
            m.simulate_bwd(packed, gimg)
        torch.cuda.synchronize()
        ts = m.timing_drain()
        fwd, bwd = sorted(ts[0::2][2:]), sorted(ts[1::2][2:])
        b1 = (8 * m.N + 8 * m.P) * wl.batch
        mf, mb = fwd[len(fwd) // 2], bwd[len(bwd) // 2]
        print(f"{wl.name} B={wl.batch} N={m.N} P={m.P} simulate pair: fwd {mf:.4f} ms + bwd {mb:.4f} ms; B1 = {b1 / 1e6:.1f} MB per "
              f"pair -> {b1 / ((mf + mb) * 1e-3) / 1e9:.1f} GB/s; fwd alone writes {4 * m.N * wl.batch / 1e6:.1f} MB -> "
              f"{4 * m.N * wl.batch / (mf * 1e-3) / 1e9:.1f} GB/s")
        return
    m.set_timing(1)
    ts = []
    for i in range(args.iters):
        if args.mode == "img":
            m.simulate_fwd(packed)
        else:
            m.loglike(packed, obs, err, None, wl.background_rms, wl.exp_time, args.mode == "grad")
        ts.append(m.last_main_ms())
    torch.cuda.synchronize()
    ts = sorted(ts[2:]) if len(ts) > 4 else ts
    print(f"{wl.name} B={wl.batch} N={m.N} P={m.P} mode={args.mode}: main kernel median {ts[len(ts)//2]:.4f} ms "
          f"min {ts[0]:.4f} ms -> {wl.batch / (ts[len(ts)//2] * 1e-3):.0f} sims/s")


if __name__ == "__main__":
    main()
