#!/usr/bin/env python
"""bench.py -- forward+grad lens simulations per second on BASELINE.json's configs[1]:
EPL+shear lens, Sersic source, 128x128 px, batch 1024, fp32, per MI355X (weak scaling over GPUs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C5|...] [--mode auto|fwdgrad|svi]

One "step" = one pass of the hot path over one batch.
  * mode fwdgrad (N = 1 default): ``ForwardProbModel.log_prob_and_grad(simulator, z)`` -- ``log_prob`` forward AND its
    gradient w.r.t. ``z`` (bijector -> fused HIP prep/main/finalize kernels -> prior), exactly what one MAP /
    HMC-leapfrog step of the reference evaluates (tf/inference.py:33-39).
  * mode svi (N > 1 default): one iteration of the sharded SVI loop (jax/inference.py:91-144) on this rank's particle
    shard: draw eps, ``z = mu + L eps`` (gl_svi_sample), the same forward+gradient call, the fused
    ``[ELBO, dELBO/dmu, dELBO/dL_packed]`` buffer (gl_svi_grad), the path's ONE collective -- an RCCL all-reduce of that
    buffer (1 + d + d(d+1)/2 floats: 105 at C2, 8 911 at C5) -- and the fused Adam launch on the surrogate's parameters
    with learning rate 0, so that every step of the measurement sees the same state.  ``inference.svi_step_buffer``
    is the product function the SVI driver itself calls.

``--gpus N`` with no launcher environment starts the N ranks itself: the parent never touches a GPU, it spawns
``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a child process and exits with its code; it refuses
(exit 2) when fewer than N devices are visible.  Under a launcher (WORLD_SIZE set) it runs as one rank.

Defaults: 1000 timed steps after 100 warm-up steps.  The chip needs a few tens of milliseconds of sustained load to settle
at its working clock, so an untimed pre-roll of 0.15 s of steps precedes the W warm-up steps whatever W and K are.

Refuses to run (exit 3) with any GIGALENS_HIP_* kernel override in the environment.  EVERY rank checks four samples of the shard
it is about to time against the float64 oracle and its last timed step for finite outputs; the flags are MIN-all-reduced and
all ranks exit 4 together on a failure anywhere.  At N = 1 the line carries a `configs` array: the other BASELINE.json configs
(C1 at batch 1 and 1024, C3 table and direct, C4, C5's per-rank shard) and the SURVEY 8f workloads C6 (catalogue members) and
C3L (linear solve), each timed for a fraction of a second with the same event ring and priced with the same two fractions (B1
bytes against HBM peak, ISA-counted flops of the dispatched kernel against the fp32 vector peak).  At N > 1 `configs` holds
BASELINE.json configs[4] itself: the 2048-particle cluster-model SVI with 2048 / N particles per rank -- forward+gradient on
the shard and the real SVI step with its 8 911-float all-reduce, timed with the same barrier / max-over-ranks protocol.

Prints ONE JSON line on rank 0 (contract in the round prompt) with two extra objects:
  roofline     -- the dominant kernel, timed by HIP-event pairs that ride on its own dispatch packets (hipExtLaunchKernel, on
                  the launch stream, no host sync) in a pass of >= 20 further steps of the same loop right after the timed
                  region -- not inside it: a timed launch costs ~5 us of its step (12.5 -> 11.8 M sims/s with every launch
                  timed), and the line's `value` must not pay for its own instrumentation --, plus the ISA-counted fp32 flop
                  rate of the dispatched instantiation against the 157.3 TFLOP/s vector peak (tools/isa_flops.py
                  disassembles the shipped code object; the binding bound of this path)
  cpu_baseline -- the oracle (reference algorithm restated op-for-op on torch-CPU, float32 + autograd) timed on 256 of
                  the workload's samples on this box's host cores: median and p10 / p90 over the timed passes
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3  # packed fp32 FMA: 256 CU x 4 SIMD x 16 lanes x 2 (packed) x 2 (FMA) x 2.4 GHz


def host_cores():
    """Host threads this process may really use: affinity, capped by the cgroup CPU quota and by the
    GPU box's per-GPU share (16), overridable with GIGALENS_CPU_THREADS."""
    if os.environ.get("GIGALENS_CPU_THREADS"):
        return max(1, int(os.environ["GIGALENS_CPU_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 16)


def _pct(sorted_vals, q):
    if not sorted_vals:
        return None
    k = min(len(sorted_vals) - 1, max(0, int(round(q * (len(sorted_vals) - 1)))))
    return sorted_vals[k]


def cpu_baseline(wl, obs, seconds=20.0, sample_batch=256, min_passes=10, max_passes=200):
    """Time the oracle (float32, torch autograd) on `sample_batch` samples of the same workload: per-pass
    forward+backward wall time; median and p10 / p90 of sims/s over the timed passes (3 warm-up passes)."""
    from oracle import ref_torch as ref
    from tests.helpers import struct_from_packed
    from gigalens_amd.model import _Packing

    cores = host_cores()
    torch.set_num_threads(cores)
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, sample_batch, dtype=torch.float32)
    x = wl.prior.sample(sample_batch, seed=11)
    packed = _Packing(wl.phys_model).pack(x, sample_batch, "cpu")
    obs_np = obs.cpu().numpy()

    def one():
        p = packed.clone().requires_grad_(True)
        ll, _ = ref.stats_pixels(rs, struct_from_packed(wl.phys_model, p), obs_np, wl.background_rms, wl.exp_time)
        ll.sum().backward()
        return p.grad

    t_w = time.perf_counter()
    one()
    first = time.perf_counter() - t_w
    n_warm = 1
    while n_warm < 3 and first * (n_warm + min_passes) < 4 * seconds:
        one()
        n_warm += 1
    ts, t0 = [], time.perf_counter()
    while len(ts) < max_passes:
        t1 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t1)
        if time.perf_counter() - t0 >= seconds and len(ts) >= min_passes:
            break
        if time.perf_counter() - t0 >= 4 * seconds:  # a very slow host: stop with what there is
            break
    rates = sorted(sample_batch / t for t in ts)
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": round(_pct(rates, 0.5), 3), "unit": "sims/s", "cores": cores, "kind": "port",
            "p10": round(_pct(rates, 0.1), 3), "p90": round(_pct(rates, 0.9), 3), "passes": len(ts),
            "warmup_passes": n_warm, "cpu_model": cpu_model,
            "sample": f"{len(ts)} timed fwd+grad passes over {sample_batch} of the {wl.batch} samples "
                      f"({wl.sim_config.num_pix}x{wl.sim_config.num_pix} px; value = median, p10 / p90 beside it; float32 "
                      f"torch-CPU restatement of the TF graph with torch.autograd, (N_pix, B) tensors as in the reference)"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs visible to this process WITHOUT touching the HIP runtime (the parent of a multi-rank run must stay GPU-free so
    that its child ranks start from a clean process): KFD topology nodes with SIMDs, narrowed by the *_VISIBLE_DEVICES
    variables.  None when the topology cannot be read (then every rank validates its own device instead)."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_ranks(args):
    """--gpus N without a launcher: start the N ranks as a CHILD process tree (the parent never initialises a GPU -- it
    counts devices from the KFD topology in sysfs, not through HIP -- and never replaces itself); exit with the child's code."""
    n_dev = visible_gpus()
    if n_dev is not None and n_dev < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible; refusing to run fewer ranks "
                         "than asked for\n")
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=dict(os.environ)))


def refuse_tuning_environment():
    """The line must describe the shipped configuration: any GIGALENS_HIP_* override (kernel selection, decomposition, the
    dissection knobs of -DGL_EXPERIMENTS builds) makes the run something else.  GIGALENS_HIP_LIB (which library file) and
    GIGALENS_DIST_BACKEND / GIGALENS_CPU_THREADS (host-side) are not overrides of the measured path."""
    bad = sorted(k for k in os.environ if k.startswith("GIGALENS_HIP_") and k != "GIGALENS_HIP_LIB")
    if bad:
        sys.stderr.write("bench.py: refusing to measure with kernel overrides in the environment: " + ", ".join(bad) + "\n")
        sys.exit(3)


EPL_SERIES_TOL = 1e-9  # csrc/gl_profiles.h epl_series_tol<float>(): where the float32 kernels stop the EPL angular series


def epl_series_stats(wl, x_struct):
    """Trips of the EPL series loop over the batch: the per-sample term count K = ceil(log(tol) / log f + 2) - 1 (capped
    at niter) with f = (1-q)/(1+q) = min(|e|, 1), restated from csrc/gl_profiles.h epl_prep (epl.py:22,37,47-54).  The
    Clenshaw loop of csrc/gl_vec.hip.h takes the terms K..0 four at a time (the table is zero-filled above K), two
    four-term groups per loop iteration with an exit after either: T = ceil((K + 1) / 4) groups, T / 2 iterations of the
    ISA-level loop.  None for models without EPL."""
    import math
    groups, ks, n = 0.0, 0.0, 0
    for prof, p in zip(wl.phys_model.lenses, x_struct.get("lens_mass", [])):
        if getattr(prof, "_kind", 0) != 1:
            continue
        e = torch.sqrt(torch.as_tensor(p["e1"], dtype=torch.float64) ** 2 + torch.as_tensor(p["e2"], dtype=torch.float64) ** 2)
        f = e.clamp(1e-30, 1.0).reshape(-1)
        cap = int(getattr(prof, "niter", 50) or 50)
        niter = math.log(EPL_SERIES_TOL) / torch.log(f) + 2.0
        K = torch.where(niter > 1, torch.ceil(niter) - 1, torch.zeros_like(niter)).clamp(max=cap)
        K = torch.where(f >= 1.0, torch.full_like(K, float(cap)), K)
        groups += float(torch.ceil((K + 1) / 4).mean())
        ks += float(K.mean())
        n += 1
    if not n:
        return None
    # the kernels peel the first four-term group out of the loop (gl_vec.hip.h four_first): the two-group loop runs the others
    return {"mean_terms": ks / n, "mean_four_term_groups": groups / n, "mean_pair_trips": 0.5 * (groups / n - 1.0), "frac_odd": 0.0}


_ISA_CACHE = {}


def isa_account(kernel_symbol, series, p_live=None, p_lens=None):
    """ISA-counted fp32 flops and VALU wave-instructions per pixel of the dispatched kernel (tools/isa_flops.py)."""
    if os.path.join(ROOT, "tools") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_flops as isa
    if "co" not in _ISA_CACHE:
        _ISA_CACHE["co"] = isa.code_object()
        _ISA_CACHE["meta"] = isa.kernel_metadata(_ISA_CACHE["co"])
    co, meta = _ISA_CACHE["co"], _ISA_CACHE["meta"]
    name = next((k for k, v in meta.items() if v["symbol"] == kernel_symbol), None)
    if name is None:
        return None
    md = meta[name]
    out = {"kernel": name, "vgpr_count": md["vgpr_count"], "vgpr_spill_count": md["vgpr_spill_count"],
           "sgpr_spill_count": md["sgpr_spill_count"], "scratch_bytes": md["scratch_bytes"]}
    model = isa.execution_model(co, name, md, series, p_live, p_lens)
    if model is not None:
        out.update(model)
    return out


def oracle_spot_check(wl, pm, sim, z, obs, err, n=4):
    """Before anything is timed: the first `n` samples of the very batch the loop will run, HIP log-likelihood against the
    float64 oracle on the same pixel grid (rtol 1e-5, BASELINE.json north_star), and a finite gradient.  The oracle is the
    checker here, never the thing measured."""
    from oracle import ref_torch as ref
    from tests.helpers import struct_from_packed
    import numpy as np
    n = max(1, min(n, z.shape[0]))
    ll = pm.log_like(sim, z).detach().double().cpu().numpy()[:n]
    lp, _, g = pm.log_prob_and_grad(sim, z)
    finite = bool(torch.isfinite(lp).all() and torch.isfinite(g).all())
    packed = sim._layout.pack(pm.bij.forward(z[:n].detach()), n, z.device).double().cpu()
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, n, dtype=torch.float64)
    with torch.no_grad():
        ll_o, _ = ref.stats_pixels(rs, struct_from_packed(wl.phys_model, packed), obs.cpu().numpy(), wl.background_rms, wl.exp_time,
                                   error_map=None if err is None else err.cpu().numpy())
    rel = float(np.max(np.abs(ll - ll_o.numpy()) / np.abs(ll_o.numpy())))
    return {"samples": n, "loglike_max_rel_err_vs_f64_oracle": float(f"{rel:.3e}"), "rtol": 1e-5,
            "logprob_and_grad_finite": finite, "ok": bool(rel <= 1e-5 and finite)}


def build_case(name, kw, workloads, ForwardProbModel, LensSimulator, dev, rank=0):
    wl = workloads.make(name, **kw)
    obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time,
                          error_map=None if err is None else err.cpu().numpy(), include_positions=False)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    from gigalens_amd import dist as gdist
    x = wl.prior.sample(wl.batch, generator=gdist.rank_generator(0, rank))
    z = pm.bij.inverse(x).to(dev).contiguous()
    return wl, obs, err, pm, sim, x, z


def roofline_of(model, wl, sim, x, kernel_ms, ms_per_step, err, stride):
    """B1 (HBM, nominal) and ISA-counted VALU fractions of the dispatched main kernel from its event-ring durations."""
    B, N, P = wl.batch, model.N, model.P
    n_planes = 1 + (1 if err is not None else 0)
    bytes_b1 = 2 * 4 * N + 2 * 4 * P
    bytes_b2 = 4 * (2 * P + 2) + 4 * N * n_planes / B
    k_mean = sum(kernel_ms) / len(kernel_ms)
    achieved = bytes_b1 * B / (k_mean * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "traffic_source": None,
            "kernel_ms": round(k_mean, 5), "kernel_ms_p10": round(_pct(kernel_ms, 0.1), 5),
            "kernel_ms_p50": round(_pct(kernel_ms, 0.5), 5), "kernel_ms_p90": round(_pct(kernel_ms, 0.9), 5),
            "kernel_launches_timed": len(kernel_ms), "kernel_event_stride": stride,
            "kernel_timing": ("HIP-event pairs on the main kernel's own dispatch packets (hipExtLaunchKernel) in a pass of further steps of "
                              "the same loop right after the timed region; not inside it: an event pair costs ~5 us of a step"),
            "kernel_share_of_step": round(k_mean / ms_per_step, 4),
            "algorithmic_bytes_per_sim_B1": bytes_b1, "algorithmic_bytes_per_sim_B2": round(bytes_b2, 1),
            "kernel_sims_per_s": round(B / (k_mean * 1e-3), 1)}
    series = epl_series_stats(wl, x)
    p_live = p_lens = None
    try:
        symbol = model.last_main_kernel()
    except Exception:  # evidence only: a model whose kernel has no name to report still gets its line
        symbol = ""
    if "gl_shp_kernel" in symbol:
        # rounds of the shapelet chains per wave-tile, counted by the kernel itself (table mode compacts a wave-tile's live pixels
        # and runs ceil(live / 64) rounds of one chain per lane: 0, 1 or 2; direct mode always 2) -- as a share of the two chains
        # per lane a tile without compaction runs, which is how the ISA model weights the chain blocks
        rows = model.partial_rows(B)
        packed = rows[:, :, 3].double()  # wave-tiles seen + 4096 x wave-tiles that ran the lens (gl_shp.hip.h: the others were
        seen = float(torch.remainder(packed, 4096.0).sum())  # provably outside the shapelet table and skipped it)
        lensed = float(torch.floor(packed / 4096.0).sum())
        p_live = 0.5 * float(rows[:, :, 2].sum()) / seen if seen > 0 else 1.0
        roof["shapelet_live_wave_tile_share"] = round(p_live, 4)
        roof["shapelet_chain_rounds_per_wave_tile"] = round(2.0 * p_live, 4)
        p_lens = lensed / seen if seen > 0 else 1.0
        roof["shapelet_wave_tiles_running_the_lens"] = round(p_lens, 4)
    n_members = sum(int(getattr(l, "n_galaxy", 0)) for l in wl.phys_model.lenses)  # galaxy catalogues (ScalingRelation / DPIESubhalo)
    try:
        # (the steady-state tile of the likelihood kernels is compiled once per variance model: tell the ISA model which ran)
        acct = isa_account(symbol, dict(series or {}, error_map=err is not None, **({"n_members": n_members} if n_members else {})), p_live, p_lens)
    except Exception as exc:  # the accounting is evidence, never a reason to lose the line
        acct = {"error": repr(exc)}
    if acct:
        roof["isa"] = acct
        fpp = acct.get("flops_per_pixel")
        if fpp:
            tflops = fpp * N * B / (k_mean * 1e-3) / 1e12
            roof["valu_flop_frac"] = round(tflops / VALU_PEAK_TFLOPS, 4)
            roof["valu_tflops"] = round(tflops, 2)
            roof["valu_peak_tflops"] = VALU_PEAK_TFLOPS
            mf = acct.get("mfma_flops_per_pixel")
            if mf:  # exact-fp32 MFMA work rides the matrix pipe (same 157.3 TFLOP/s peak), reported beside the VALU figure
                roof["mfma_tflops"] = round(mf * N * B / (k_mean * 1e-3) / 1e12, 2)
    return roof, series


def timed_steps(step, model, steps, warmup, preroll_s, stride, sync_barrier=None):
    """Pre-roll, W warm-up steps, K timed steps (no events inside), then the kernel pass (see kernel_pass)."""
    t_pre, n_pre = time.perf_counter(), 0
    while time.perf_counter() - t_pre < preroll_s:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n_pre += 20
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = kernel_pass(step, model, max(20, min(steps, 100)), stride)
    return elapsed, kernel_ms, n_pre


def all_ranks_ok(ok, dev):
    """MIN over ranks of a validity flag: every rank learns whether ANY rank failed, so that all exit together (a rank leaving
    on its own would strand the others in the next collective until the launcher kills them)."""
    from gigalens_amd import dist as gdist
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    gdist.allreduce_min_(flag)
    return bool(flag.item() > 0.5)


def make_svi_step(wl, pm, sim, B, dev, rank):
    """The real SVI iteration on this rank's particle shard (inference.svi_step_buffer + the fused Adam launch with learning
    rate 0, so that every step of the measurement sees the same state).  Surrogate state: the reference's SVI start
    (tf/inference.py:47-72: mean = a MAP-like point, scale 1e-3 I); the mean is the prior draw whose EPL series length is closest
    to the batch mean, so that a particle costs what an average sample costs (identical on every rank: rank 0's stream).
    Returns (step, series of the surrogate's mean or None)."""
    import math
    from gigalens_amd import dist as gdist
    from gigalens_amd import inference as ginf
    x0 = wl.prior.sample(B, generator=gdist.rank_generator(0, 0))
    z0 = pm.bij.inverse(x0)
    d = z0.shape[1]
    pick, series = 0, None
    s0 = epl_series_stats(wl, x0)
    if s0 is not None:
        e = None
        for prof, p in zip(wl.phys_model.lenses, x0["lens_mass"]):
            if getattr(prof, "_kind", 0) == 1:
                e = torch.sqrt(torch.as_tensor(p["e1"], dtype=torch.float64) ** 2 + torch.as_tensor(p["e2"], dtype=torch.float64) ** 2)
                break
        K = torch.ceil(math.log(EPL_SERIES_TOL) / torch.log(e.clamp(1e-30, 1 - 1e-12)) + 2.0) - 1
        pick = int(torch.argmin((K - round(s0["mean_terms"])).abs()))
        g = float(torch.ceil((K[pick] + 1) / 4))
        series = {"mean_terms": float(K[pick]), "mean_four_term_groups": g, "mean_pair_trips": 0.5 * (g - 1.0), "frac_odd": 0.0}
    mu = z0[pick].to(dev).contiguous().clone()
    lpk = ginf.tril_pack(torch.eye(d, device=dev) * 1e-3)
    sv_params = torch.cat([mu, lpk]).contiguous()
    opt = ginf.Adam(lr=0.0)
    gen = gdist.rank_generator(2, rank, device=dev)

    def vg(zz):
        lp_, _, g_ = pm.log_prob_and_grad(sim, zz)
        return lp_, g_

    pool = ginf.NormalPool(gen, B, d, device=dev)  # the driver's own draw schedule (ModellingSequence.SVI)

    def step():
        buf = ginf.svi_step_buffer(sv_params[:d], sv_params[d:], None, B, gen, value_and_grad_fn=vg, full_rank=True, eps=pool.next())
        opt.step(sv_params, buf[1:])
        return buf

    return step, series


def kernel_pass(step, model, n, stride=1):
    """The main kernel's launch durations from HIP events: `n` further steps of the very loop that was just timed, same stream,
    same clock state, every stride-th launch carrying an event pair (sorted ms).  A pass of its own because an event pair is not
    free: on the dispatch packet (hipExtLaunchKernel) it costs ~5 us per step, as two event records ~2.5 us each -- inside the
    timed region it would lower the very `value` the line is about (12.5 -> 11.8 M sims/s at C2 with every launch timed)."""
    n_ev = (n + stride - 1) // stride
    model.set_timing(n_ev, stride)
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = sorted(model.timing_drain())
    model.set_timing(0)
    return ms


def timed_region(step, model, steps, warmup, events, stride, dev):
    """W warm-up steps, then K steps between barrier + synchronize pairs and the MAX over ranks of the elapsed time (no events
    inside: see kernel_pass); then the kernel pass -- max(20, min(K, 200)) more steps with their main launches timed.  Returns
    (elapsed, sorted kernel ms, last timed step's outputs)."""
    from gigalens_amd import dist as gdist
    for _ in range(warmup):
        step()
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    gdist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    gdist.allreduce_max_(tmax)
    kernel_ms = kernel_pass(step, model, max(20, min(steps, 200)), stride) if events else []
    return float(tmax.item()), kernel_ms, last


def outputs_finite(last):
    outs = last if isinstance(last, (tuple, list)) else (last,)
    return all(bool(torch.isfinite(t).all()) for t in outs if torch.is_tensor(t))


def measure_c5_sharded(dev, rank, world, steps, warmup, events, stride):
    """BASELINE.json configs[4] on the line of an N > 1 run: the 2048-particle cluster-model SVI, 2048 / N particles per rank --
    forward+gradient on the shard (no collective) and the real SVI iteration with its 8 911-float all-reduce, each for `steps`
    steps with the same barrier / max-over-ranks timing as the headline.  Every rank checks its own shard; the flags are
    reduced by the caller."""
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    total = 2048
    B = max(1, total // world)
    wl, obs, err, pm, sim, x, z = build_case("C5", dict(batch=B), workloads, ForwardProbModel, LensSimulator, dev, rank)
    model = sim._model
    d = z.shape[1]
    check = oracle_spot_check(wl, pm, sim, z, obs, err, n=2)
    ok = check["ok"]
    e_fg, k_fg, last = timed_region(lambda: pm.log_prob_and_grad(sim, z), model, steps, warmup, events, stride, dev)
    ok = ok and outputs_finite(last)
    svi, _ = make_svi_step(wl, pm, sim, B, dev, rank)
    e_svi, k_svi, last = timed_region(svi, model, steps, warmup, events, stride, dev)
    ok = ok and outputs_finite(last)
    ms_fg = 1e3 * e_fg / steps
    out = {"config": "BASELINE.json configs[4]: full SVI, 2048 particles, cluster model (8 NFW + 20 Sersic) 256x256 px, particles sharded "
                     f"over {world} ranks",
           "workload": f"{wl.name}: {wl.description}", "particles_total": B * world, "particles_per_rank": B, "pixels": model.N,
           "z_dim": d, "steps": steps, "warmup": warmup,
           "fwdgrad": {"ms_per_step": round(ms_fg, 4), "sims_per_s": round(B * world * steps / e_fg, 1),
                       "what": "ForwardProbModel.log_prob_and_grad on every rank's shard, no collective"},
           "svi_step": {"ms_per_step": round(1e3 * e_svi / steps, 4), "particles_per_s": round(B * world * steps / e_svi, 1),
                        "allreduce_floats": 1 + d + d * (d + 1) // 2, "backend": torch.distributed.get_backend() if world > 1 else None,
                        "what": ("inference.svi_step_buffer (eps draw, gl_svi_sample, log_prob forward+gradient, gl_svi_grad, all-reduce of "
                                 "the fused [ELBO, grad] buffer) + fused Adam launch (lr 0)")},
           "oracle_spot_check_rank0": check if rank == 0 else None}
    if k_fg:
        roof, _ = roofline_of(model, wl, sim, x, k_fg, ms_fg, err, stride)
        out.update({"kernel": roof.get("isa", {}).get("kernel"), "kernel_ms": roof["kernel_ms"], "hbm_frac_B1": roof["frac"],
                    "valu_flop_frac": roof.get("valu_flop_frac"), "valu_insts_per_pixel": roof.get("isa", {}).get("valu_insts_per_pixel")})
    del pm, sim, z
    torch.cuda.empty_cache()
    return out, ok


EXTRA_CONFIGS = [  # the other BASELINE.json configs, measured after the headline line's loop (N = 1 only), a fraction of a second each
    ("C1", dict(), "BASELINE.json configs[0]: SIE lens + Sersic source, 64x64 px, batch 1"),
    ("C1", dict(batch=1024), "configs[0] model at batch 1024"),
    ("C3", dict(interpolate=True), "BASELINE.json configs[2]: Shapelets n_max=10 source, 128x128 px, batch 1024 (table mode, the reference's default)"),
    ("C3", dict(interpolate=False), "configs[2], direct (Hermite recurrence) mode"),
    ("C4", dict(), "BASELINE.json configs[3]: 8 NFW halos + 20 Sersic sources, 256x256 px, batch 512"),
    ("C5", dict(), "BASELINE.json configs[4]: per-rank shard (256 particles) of the 2048-particle cluster-model SVI, forward+gradient"),
    ("C6", dict(), "SURVEY 8f-3 (the cluster-lens workload): dPIE halo + 200 catalogue member galaxies (DPIESubhalo) + 20 Sersic sources, "
                   "256x256 px, batch 128"),
]


def measure_c3l(dev, seconds=0.25):
    """the linear amplitude solve (lstsq_simulate) at C3L -- 1024 samples, 128 x 128 px, one shapelet source
    n_max = 10 solved linearly (66 + 1 channels): ms per gl_lstsq_fwd (normal matrix straight from the bases + Cholesky
    attempt), HIP-event timed."""
    from gigalens_amd import workloads
    from gigalens_amd.simulator import LensSimulator
    try:
        wl = workloads.make("C3L")
        c2 = workloads.make("C2", num_pix=wl.sim_config.num_pix, batch=1)
        obs, _, _ = workloads.synthetic_observation(c2, LensSimulator)
        errm = torch.sqrt(wl.background_rms ** 2 + obs.clamp_min(0) / wl.exp_time).contiguous()
        sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
        packed = sim.pack(wl.prior.sample(wl.batch, seed=0)).contiguous()
        m = sim._model
        for _ in range(3):
            (coeffs,) = m.lstsq(packed, obs, errm, 7, want="coeffs")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.lstsq(packed, obs, errm, 7, want="coeffs")
        torch.cuda.synchronize()
        iters = int(min(400, max(20, seconds / max(time.perf_counter() - t0, 1e-5))))
        e0.record()
        for _ in range(iters):
            (coeffs,) = m.lstsq(packed, obs, errm, 7, want="coeffs")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        return {"config": "SURVEY 8f-4: lstsq_simulate at C3L (1024 samples, 128x128 px, shapelets n_max = 10 solved linearly, 66 + 1 channels)",
                "workload": f"{wl.name}: {wl.description}", "batch": wl.batch, "pixels": m.N, "linear_channels": m.num_linear(),
                "steps": iters, "ms_per_step": round(ms, 4), "solves_per_s": round(wl.batch / (ms * 1e-3), 1),
                "coefficients_finite": bool(torch.isfinite(coeffs).all()),
                "what": "gl_lstsq_fwd (coefficients): front end, gl_shp_normal_kernel (exact-fp32 MFMA normal matrix from the bases), "
                        "gl_chol_solve_kernel; timed end to end with HIP events"}
    except Exception as exc:
        return {"config": "SURVEY 8f-4: lstsq_simulate at C3L", "error": repr(exc)}


def measure_extra_configs(dev, steps_cap=200):
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    out = []
    for name, kw, what in EXTRA_CONFIGS:
        try:
            wl, obs, err, pm, sim, x, z = build_case(name, kw, workloads, ForwardProbModel, LensSimulator, dev)
            check = oracle_spot_check(wl, pm, sim, z, obs, err, n=2 if sim._model.N > 16384 else 4)
            probe_t = time.perf_counter()
            for _ in range(3):
                pm.log_prob_and_grad(sim, z)
            torch.cuda.synchronize()
            per = max((time.perf_counter() - probe_t) / 3, 1e-5)
            steps = int(min(steps_cap, max(20, 0.25 / per)))
            elapsed, kernel_ms, _ = timed_steps(lambda: pm.log_prob_and_grad(sim, z), sim._model, steps, max(5, steps // 10), 0.05, 1)
            # two more timed loops of the same length; the entry carries the MEDIAN of the three (a shared box now and then
            # stalls one loop by tens of per cent -- seen once in ten runs on the C3 entry; the headline `value` above is exactly
            # the K steps the contract names, never a pick)
            runs = [elapsed]
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    pm.log_prob_and_grad(sim, z)
                torch.cuda.synchronize()
                runs.append(time.perf_counter() - t0)
            elapsed = sorted(runs)[1]
            ms = 1e3 * elapsed / steps
            roof, series = roofline_of(sim._model, wl, sim, x, kernel_ms, ms, err, 1)
            out.append({"config": what, "workload": f"{wl.name}: {wl.description}", "batch": wl.batch, "pixels": sim._model.N,
                        "params_per_sample": sim._model.P, "steps": steps, "ms_per_step": round(ms, 4),
                        "ms_per_step_of_three_loops": [round(1e3 * r / steps, 4) for r in runs],
                        "sims_per_s": round(wl.batch * steps / elapsed, 1), "kernel": roof.get("isa", {}).get("kernel"),
                        "kernel_ms": roof["kernel_ms"], "hbm_frac_B1": roof["frac"], "valu_flop_frac": roof.get("valu_flop_frac"),
                        "valu_insts_per_pixel": roof.get("isa", {}).get("valu_insts_per_pixel"),
                        "flops_per_pixel": roof.get("isa", {}).get("flops_per_pixel"),
                        "shapelet_live_wave_tile_share": roof.get("shapelet_live_wave_tile_share"),
                        "shapelet_wave_tiles_running_the_lens": roof.get("shapelet_wave_tiles_running_the_lens"),
                        "epl_series": series, "oracle_spot_check": check})
            if wl.batch * sim._model.N <= 300_000:
                # host-issue bound sizes: the same call with graph=True (opt-in: the launch sequence replayed from a HIP graph,
                # static outputs -- model.py log_prob_and_grad), z being the graph's own input; same results checked first
                try:
                    zs = pm.graph_input(sim, z)
                    ref_out = pm.log_prob_and_grad(sim, z)
                    got = pm.log_prob_and_grad(sim, zs, graph=True)
                    same = all(bool(torch.equal(a, b)) for a, b in zip(ref_out, got))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        pm.log_prob_and_grad(sim, zs, graph=True)
                    torch.cuda.synchronize()
                    out[-1]["graph_replay"] = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / steps, 4), "steps": steps,
                                               "equals_stream_launches": same,
                                               "what": "log_prob_and_grad(sim, z, graph=True): opt-in, not the default path the line above times"}
                except Exception as exc:
                    out[-1]["graph_replay"] = {"error": repr(exc)}
            del pm, sim, z
            torch.cuda.empty_cache()
        except Exception as exc:  # a config that fails must not cost the headline line
            out.append({"config": what, "error": repr(exc)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--mode", default="auto", choices=["auto", "fwdgrad", "svi"],
                    help="auto: fwdgrad on one GPU, svi (sharded particles + the all-reduce) on several")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the workload's)")
    ap.add_argument("--num-pix", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (the `configs` array of the line)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="skip the kernel pass after the timed region (roofline is then null)")
    ap.add_argument("--kernel-event-stride", type=int, default=1,
                    help="time every k-th main-kernel launch of the kernel pass that follows the timed region (event pairs on the "
                         "kernels' own dispatch packets)")
    ap.add_argument("--preroll-seconds", type=float, default=0.15,
                    help="untimed sustained load before the warm-up steps (clock settling); 0 disables")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-samples", type=int, default=None)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    refuse_tuning_environment()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)  # does not return

    from gigalens_amd import dist as gdist
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} rank(s)\n")
        sys.exit(2)
    if torch.cuda.device_count() < 1:
        sys.stderr.write("bench.py: no GPU visible (gigalens_amd has no CPU path)\n")
        sys.exit(2)
    rank, local_rank, world = gdist.init_from_env()
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import __graft_entry__ as ge
    from gigalens_amd import _native, workloads
    from gigalens_amd import inference as ginf
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    if not os.path.exists(_native.lib_path()):
        ge.build()

    if world > 1 and torch.distributed.get_backend() == "nccl" and torch.cuda.device_count() < world:
        # (the parent's sysfs count can exceed what a device cgroup lets the ranks open: then every rank says so and leaves)
        sys.stderr.write(f"bench.py: rank {rank}: {world} ranks but only {torch.cuda.device_count()} GPU(s) can be opened\n")
        sys.exit(2)
    if world > 1:  # bring the communicator up (RCCL builds its rings / trees on the first collective: seconds) before anything is timed
        warm = torch.ones(8, device=dev)
        gdist.allreduce_mean_(warm)
        gdist.barrier()
        torch.cuda.synchronize()
    mode = args.mode if args.mode != "auto" else ("svi" if world > 1 else "fwdgrad")
    wl, obs, err, pm, sim, x, z = build_case(args.workload, dict(num_pix=args.num_pix, batch=args.batch), workloads,
                                             ForwardProbModel, LensSimulator, dev, rank)
    model = sim._model
    B, N, P = wl.batch, model.N, model.P
    series = epl_series_stats(wl, x)
    d = z.shape[1]
    n_coll = 1 + d + d * (d + 1) // 2
    # the batch the loop will run, checked against the float64 oracle BEFORE anything is timed -- on EVERY rank, each on its own
    # shard (4 samples; 2 at the 256 x 256 configs); the flags are reduced with a MIN all-reduce and a wrong or non-finite
    # result on any rank ends the run on all of them: a line is never printed for numbers nobody looked at
    check = oracle_spot_check(wl, pm, sim, z, obs, err, n=2 if N > 16384 else 4)
    if not all_ranks_ok(check["ok"], dev):
        sys.stderr.write(f"bench.py: rank {rank}: the timed batch does not match the oracle on some rank (this rank: {check})\n")
        sys.exit(4)

    if mode == "svi":
        step, series_svi = make_svi_step(wl, pm, sim, B, dev, rank)
        if series_svi is not None:
            series = series_svi
    else:
        def step():
            return pm.log_prob_and_grad(sim, z)

    # untimed pre-roll: sustained load until the chip has settled at its working clock (see the module docstring), so
    # that the timed region reads the same whatever W and K the caller picks; then the W warm-up steps of the contract
    # (the pre-roll is timed per rank, so it must not contain a collective: ranks would disagree on the number of calls --
    # it runs the collective-free forward+gradient call, which is the load that matters for the clock)
    t_pre = time.perf_counter()
    n_pre = 0
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(50):
            pm.log_prob_and_grad(sim, z)
        torch.cuda.synchronize()
        n_pre += 50
    events = not args.no_kernel_events
    # the kernel pass after the timed region times every main launch (event pairs on the dispatch packets, csrc/gl_launch.hip.h);
    # --kernel-event-stride thins it out if asked to
    stride = max(1, min(args.kernel_event_stride, args.steps))
    elapsed, kernel_ms, last = timed_region(step, model, args.steps, args.warmup, events, stride, dev)
    # the last timed step's outputs must be finite numbers on every rank (a NaN or an untouched buffer would print the same line)
    finite_after = outputs_finite(last)
    # beside the line's value (not part of it): with several ranks, the same K steps of the plain forward+gradient call on
    # each rank's shard and NO collective -- how MAP and HMC shard (jax/inference.py:32-80,157-208) -- so that the cost of the
    # SVI step's extra launches and of its all-reduce can be read off the line
    sharded = None
    kernel_ms_fg = []
    if world > 1 and mode == "svi":
        e2, kernel_ms_fg, last2 = timed_region(lambda: pm.log_prob_and_grad(sim, z), model, args.steps, min(args.warmup, 20), events,
                                               stride, dev)
        finite_after = finite_after and outputs_finite(last2)
        sharded = {"value": round(B * world * args.steps / e2, 1), "unit": "sims/s", "ms_per_step": round(1e3 * e2 / args.steps, 4),
                   "what": "ForwardProbModel.log_prob_and_grad on every rank's shard, no collective (MAP / HMC sharding)"}
    # BASELINE.json configs[4] rides on the N > 1 line: the 2048-particle cluster-model SVI sharded over the ranks
    c5 = None
    c5_ok = True
    if world > 1 and args.mode == "auto" and args.workload.upper() == "C2" and not args.no_configs:
        c5, c5_ok = measure_c5_sharded(dev, rank, world, min(args.steps, 200), min(max(args.warmup, 5), 20), events, stride)
    if not all_ranks_ok(finite_after and c5_ok, dev):
        sys.stderr.write(f"bench.py: rank {rank}: a timed loop returned non-finite values or the C5 shard failed its check on some rank "
                         f"(this rank: finite {finite_after}, C5 ok {c5_ok})\n")
        sys.exit(4)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        sims = B * world * args.steps / elapsed
        # With several ranks and no --mode given, the line's `value` stays the metric BASELINE.json names at every N -- the
        # forward+gradient step on every rank's shard of the samples, which has no data-path collective -- so that the per-N
        # values compare like with like; the SVI driver's step, the one place the path exchanges data (one all-reduce of the
        # fused [ELBO, gradient] buffer per step), is timed in the same run and reported beside it as `svi_step`.
        svi_side = None
        line_mode = mode
        if world > 1 and args.mode == "auto" and sharded is not None:
            svi_side = {"value": round(sims, 1), "unit": "particles/s", "ms_per_step": round(ms_per_step, 4),
                        "allreduce_floats": n_coll, "backend": torch.distributed.get_backend(),
                        "what": ("inference.svi_step_buffer (eps draw, gl_svi_sample, log_prob forward+gradient, gl_svi_grad, "
                                 "all-reduce of the fused [ELBO, grad] buffer) + fused Adam launch (lr 0), particles sharded")}
            sims, ms_per_step, line_mode = sharded["value"], sharded["ms_per_step"], "fwdgrad"
            kernel_ms = kernel_ms_fg or kernel_ms
        roofline = None
        if kernel_ms:
            roofline, series_fg = roofline_of(model, wl, sim, x, kernel_ms, ms_per_step, err, stride)
            if line_mode != "svi":
                series = series_fg
            roofline["note"] = ("path is VALU/transcendental-bound (SURVEY 8d): the HBM fraction is priced with the "
                                "simulate()-boundary bytes B1 as the metric asks; the binding bound is the fp32 vector rate "
                                "(valu_flop_frac)")
            try:
                # HBM bytes per launch of this kernel are a PMC figure (2 x FETCH_SIZE + WRITE_SIZE, separate counter passes of
                # tools/collect_profiles.sh): the committed summary of the newest profiled run of the same workload, named beside it
                prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_summary.json"))
                if prof and args.workload.upper() == "C2" and B == 1024:
                    summ = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
                    t = summ.get("traffic_bytes_per_launch")
                    if t:
                        roofline["traffic"] = round(float(t), 1)
                        roofline["traffic_source"] = (f"profiles/{prof[-1]}::traffic_bytes_per_launch (rocprofv3 --pmc FETCH_SIZE / "
                                                      "WRITE_SIZE passes of tools/collect_profiles.sh on tools/prof_kernel.py, same "
                                                      "workload and kernel; not a measurement of this run)")
                        roofline["traffic_over_B2"] = round(float(t) / (roofline["algorithmic_bytes_per_sim_B2"] * B), 2)
                        roofline["traffic_over_B1"] = round(float(t) / (roofline["algorithmic_bytes_per_sim_B1"] * B), 4)
            except Exception:
                pass
        out = {
            "metric": "forward+grad lens sims/sec, 128x128 px batch 1024; achieved HBM GB/s vs peak",
            "value": round(sims, 1), "unit": "sims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl.name}: {wl.description}, {wl.sim_config.num_pix}x{wl.sim_config.num_pix} px, "
                                   f"batch {B} per GPU, fp32" + (" (BASELINE.json configs[1])" if wl.name == "C2" else "")
                                   + (" (BASELINE.json configs[4]: per-rank shard of the 2048-particle SVI)" if wl.name == "C5" else ""),
                       "samples_per_gpu": B, "pixels": N, "params_per_sample": P, "z_dim": d, "mode": line_mode,
                       "untimed_preroll_steps": n_pre, "kernel_events_in_timed_loop": False, "kernel_pass_after_timed_loop": events,
                       "epl_series": series,
                       "parallelism": ("single GPU" if world == 1 else
                                       (f"dp{world}: particle shards, one {n_coll}-float all-reduce per step "
                                        f"(torch.distributed backend {torch.distributed.get_backend()}; nccl = RCCL over xGMI)"
                                        if line_mode == "svi" else
                                        f"dp{world}: samples sharded over the ranks, no data-path collective in this step (barrier + "
                                        f"max-over-ranks timing only); the SVI step with its {n_coll}-float all-reduce "
                                        f"(backend {torch.distributed.get_backend()}; nccl = RCCL over xGMI) is `svi_step`")),
                       "step": ("inference.svi_step_buffer (eps draw, gl_svi_sample, log_prob forward+gradient, gl_svi_grad, all-reduce of "
                                f"the fused {n_coll}-float [ELBO, grad] buffer) + fused Adam launch (lr 0)" if line_mode == "svi" else
                                "ForwardProbModel.log_prob_and_grad: log_prob forward + gradient w.r.t. z (bijector, "
                                "kernels, prior) in one native launch sequence")},
            "roofline": roofline,
            "oracle_spot_check": check,
        }
        out["oracle_spot_check"] = dict(check, ranks_checked=world, all_ranks_ok=True)
        if svi_side is not None:
            out["svi_step"] = svi_side
        elif sharded is not None:
            out["sharded_fwdgrad_without_collective"] = sharded
        if c5 is not None:
            out["configs"] = [c5]
        if world > 1:
            out["multi_gpu_note"] = (f"backend {torch.distributed.get_backend()}: until a SCALE_rNN.json of the driver exists, no N > 1 line "
                                     "of this repository has been measured over RCCL / xGMI (the builder's boxes have one GPU; rehearsals "
                                     "run two gloo ranks on it, profiles/*_2rank_gloo_rehearsal.json)")
        if world == 1 and not args.no_configs and args.workload.upper() == "C2":
            del pm, sim, z
            torch.cuda.empty_cache()
            out["configs"] = measure_extra_configs(dev)  # the other BASELINE configs, each with both fractions
            out["configs"].append(measure_c3l(dev))
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            n_cpu = args.cpu_samples or (256 if N <= 16384 else 32)
            out["cpu_baseline"] = cpu_baseline(wl, obs, seconds=args.cpu_seconds, sample_batch=min(n_cpu, B))
        print(json.dumps(out), flush=True)
    gdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
