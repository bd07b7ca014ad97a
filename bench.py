#!/usr/bin/env python3
"""Benchmark of the batched NMPC hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,4,5}]
                    [--instances B] [--horizon H]

--config selects the workload (the numbered configurations of the reference
benchmark definition):

  2 (default)  B = 4096 independent controllers, horizon 100.  One step = one
               cold solve of the batch: the reference's start-up sequence per
               controller (state reset, nmpc_init, wind, H+1 reference points
               -- which runs the RK4 + finite-difference linearisation of every
               horizon stage -- one feedback step, get controls), all inputs
               resident in HBM.  --instances 32768 is the per-GPU share of the
               262 144-instance, 8-GPU configuration.
  4            the same step for 1024 controllers with a 500-stage horizon.
  5            2^20 independent (x, u, wind) samples through the stage
               linearisation only (16 RK4 integrations + Jacobian per sample).
  1            one controller through the reference's 24-symbol interface:
               latency of the 50 Hz control step, beside the CPU on one core.

With --gpus N > 1 (launched through torch.distributed.run) every rank owns B
instances on its own GPU (weak scaling) and the per-step results are
all-gathered over RCCL.

Prints one JSON line on rank 0.  Besides the whole-job rate it carries
  roofline      the dominant kernel's algorithmic HBM bytes per launch over its
                mean launch duration, measured with HIP events on the stream
                the kernels run on, during the timed steps; next to the vendor
                peaks the peaks measured in the same run (stream triad, fp64
                vector rate with and without fused multiply-add)
  cpu_baseline  the CPU oracle port (oracle/nmpc_port.c, a restatement of the
                reference's pure-C path, bit-identical to it) or, for config 1,
                the compiled reference itself where it is present, on the same
                inputs on the host cores of this box.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# --precision float measures libcnmpc_f32.so (real_t = float, the reference's
# shipped configuration); the library is chosen before nmpc_amd is imported
if "--precision" in sys.argv and sys.argv[sys.argv.index("--precision") + 1:][:1] == ["float"]:
    os.environ["NMPC_AMD_LIB"] = os.path.join(ROOT, "nmpc_amd", "libcnmpc_f32.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before libcnmpc.so: see nmpc_amd/_lib.py)

from nmpc_amd import _lib  # noqa: E402
from nmpc_amd import synthetic as S  # noqa: E402
from nmpc_amd.batch import Batch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, spec
FP64_VECTOR_PEAK_TF = 78.6  # MI355X fp64 vector, spec (FMA counted as 2 flop)
FP32_VECTOR_PEAK_TF = 157.3 # MI355X fp32 vector, spec
# arithmetic type of the loaded library: set in main()
REAL = dict(name="f64", np=np.float64, torch=None, bytes=8, precision="double", peak_tf=FP64_VECTOR_PEAK_TF)
METRIC = "NMPC solves/sec (prep+feedback) at batch B, default horizon"

# Algorithmic work per unit, from the operation counts of the reference
# functions (one flop per add/sub/mul/div/sqrt/compare-select):
#   linearise     per stage: 16 RK4 integrations + 16 state differences;
#                 reads x_k, u_k (128 B), writes C (1440 B) and c (96 B)
#   newton_solve  per stage block and Newton iteration: two substitutions
#                 (0.84 kflop) that read the factor block row once each, on the
#                 way down and on the way up (2 x 1824 B: 228 values, the
#                 sub-diagonal block and the lower triangle; 180 kB of factor per
#                 instance do not stay on chip between the sweeps), the
#                 gradient (96 B) and read/write the step (3 x 96 B): 4032 B;
#                 blocks that are refactorised add the reverse Cholesky of a
#                 12x12 diagonal + 12x12 sub-diagonal block (4.2 kflop), read the
#                 Newton Hessian's block row (the same 228 values) instead of the
#                 factor's on the way down and store the factor: 1824 B more.  Every block is refactorised in
#                 the first iteration; later iterations refactorise from the
#                 last active-set change down, which is not counted here.
#   newton_setup  per stage and iteration: gradient (0.38 kflop) + Hessian
#                 blocks (2.9 kflop: the lower triangle of the diagonal block, 84 entries of up
#                 to 15 dyads, and the sub-diagonal block); reads C, c, z, y (1896 B), writes the
#                 block row (228 values) and gradient (1920 B)
#   newton_step / qp_begin  per stage and call: stage update, clipping,
#                 objective; reads C, c and the 15-vectors, writes them back
ALGO = {
    "linearise": dict(flop=17000.0, bytes=1664.0),
    "newton_solve": dict(flop=840.0, bytes=4032.0),
    "newton_factor": dict(flop=4200.0, bytes=1824.0),
    "newton_setup": dict(flop=3300.0, bytes=3816.0),
    "newton_step": dict(flop=670.0, bytes=3100.0),
    "qp_begin": dict(flop=260.0, bytes=1100.0),
}
# what limits each kernel: linearise sits at the fp64-vector / HBM ridge and is
# instruction-bound; the Newton solve is a dependent chain per instance,
# HBM-bound only while every block is factorised
BOUND = {"linearise": "vector-alu", "newton_solve": "hbm", "qp_begin": "hbm", "newton_step": "hbm"}
LIMITER = {
    "linearise": "vector instruction issue (FMA contraction is off: the ceiling is the FMA-free rate)",
    "newton_solve": "instruction issue of one wavefront per SIMD running one dependent chain per "
                    "instance (long sweeps: two wavefronts, factorisation and substitution, still one "
                    "chain; from 512 instances on an iteration's solve is two overlapping "
                    "launches, long and short factorisation sweeps, each counted in full); HBM in the "
                    "launch that factorises every block and for batches that fill the chip twice over",
    "qp_begin": "HBM", "newton_step": "HBM",
}



def same_bits(a, b):
    """Same shape and, element for element, the same bit pattern (-0.0 is not +0.0);
    two NaNs count as equal (x86 and gfx950 default NaNs differ in sign)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind != "f" or a.dtype != b.dtype:
        return bool(np.array_equal(a, b))
    bits = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return bool(np.all((a.view(bits) == b.view(bits)) | (np.isnan(a) & np.isnan(b))))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 4, 5))
    ap.add_argument("--instances", type=int, default=None, help="controller instances per GPU")
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--samples", type=int, default=1 << 20, help="config 5: samples per GPU")
    ap.add_argument("--seed", type=int, default=S.DEFAULT_SEED)
    ap.add_argument("--warm-steps", type=int, default=20,
                    help="warm control steps timed after the cold solves (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-peaks", action="store_true", help="skip the peak microbenchmarks")
    ap.add_argument("--precision", choices=("double", "float"), default="double",
                    help="float: measure libcnmpc_f32.so (configs 2, 4, 5)")
    ap.add_argument("--cpu-threads", type=int, default=None,
                    help="threads of the CPU baseline (default: every core of this job's share)")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="units of the workload the CPU baseline processes per pass")
    a = ap.parse_args()
    if a.instances is None:
        a.instances = {1: 1, 2: 4096, 4: 1024, 5: 0}[a.config]
    if a.horizon is None:
        a.horizon = 500 if a.config == 4 else 100
    if a.steps is None:
        a.steps = {1: 200, 2: 100, 4: 40, 5: 100}[a.config]      # enough samples for a p99 that is not the maximum
    return a


def host_cores(requested=None):
    """Threads the CPU baseline runs on, and what the host offers.

    The baseline uses every core this process may use: the scheduler affinity,
    cut to the job's CPU share where the launcher states one (OMP_NUM_THREADS:
    a one-GPU share of a many-GPU host exports its core count that way while the
    affinity mask still shows the whole machine).  --cpu-threads overrides."""
    avail = len(os.sched_getaffinity(0))
    share = None
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", ""))
    except ValueError:
        pass
    threads = requested or (min(avail, share) if share and share > 0 else avail)
    model = None
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return max(1, threads), dict(cores_available=avail, cpu_share=share, cpu_model=model)


def percentile(sorted_ms, q):
    return sorted_ms[min(len(sorted_ms) - 1, int(q * len(sorted_ms)))]


def measure_peaks(device):
    """Stream triad and fp64 vector rates of this GPU, measured now."""
    lib = _lib.load()
    out = {}
    v = C.c_double()
    if lib.nmpc_measure_stream_triad(device, 1 << 30, C.byref(v)) == 0:
        out["hbm_triad_GBps"] = round(v.value, 1)
    if lib.nmpc_measure_fp64_vector_peak(device, 0, C.byref(v)) == 0:
        out["fp64_vector_no_fma_TFLOPs"] = round(v.value, 2)
    if lib.nmpc_measure_fp64_vector_peak(device, 1, C.byref(v)) == 0:
        out["fp64_vector_fma_TFLOPs"] = round(v.value, 2)
    return out


def traffic_file(tag):
    """The PMC traffic summary of this workload: profiles/traffic_<tag>.json, which
    tools/pmc_traffic.py rewrites whenever the counters are collected again."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % tag)
    return path if os.path.exists(path) else None


def traffic_source(path):
    """File name plus what pins its contents: the library sources' hash recorded in it
    when it was collected, and whether that still is the hash of the sources here."""
    rel = os.path.relpath(path, ROOT)
    j = json.load(open(path))
    then = j.get("kernel_sources_sha1")
    now = kernel_sources_sha1()
    return dict(file=rel, collected_on_sources=then, sources_now=now,
                stale=(then != now) if then else None)


def kernel_sources_sha1():
    """One hash over the library's kernel sources, the thing PMC counters go stale against."""
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(ROOT, "nmpc_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:12]


def load_traffic(tag, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this
    workload (tools/pmc_traffic.py), or None."""
    path = traffic_file(tag)
    if REAL["name"] != "f64":       # the counters were collected on the double-precision library
        return None
    if path:
        return json.load(open(path)).get("bytes_per_launch", {}).get(kernel)
    return None


def whole_step_traffic(tag, ms_per_step, peaks):
    """HBM bytes of one whole step from the committed PMC summary (bytes per launch
    x launches per step of every kernel in it) over the measured step time: what
    the path as a whole takes from the memory system, next to the one kernel the
    roofline entry prices."""
    path = traffic_file(tag)
    if REAL["name"] != "f64" or not path:
        return None
    j = json.load(open(path))
    per, n = j.get("bytes_per_launch", {}), j.get("launches", {})
    steps = j.get("steps") or max(1, min(n.get("qp_begin", 1), n.get("linearise", 1)))
    total = sum(per[k] * n.get(k, 0) for k in per) / steps
    gbps = total / 1e9 / (ms_per_step / 1e3)
    out = dict(traffic=round(total), GBps=round(gbps, 1), frac=round(gbps / HBM_PEAK_GBS, 5),
               source=traffic_source(path))
    if peaks and peaks.get("hbm_triad_GBps"):
        out["frac_of_measured_hbm"] = round(gbps / peaks["hbm_triad_GBps"], 5)
    return out


def make_roofline(kernel, kern, algo_bytes, peaks, traffic):
    k = kern[kernel]
    achieved = k["algo_GBps"]
    r = dict(kernel=kernel, bound=BOUND.get(kernel, "hbm"), achieved=achieved, peak=HBM_PEAK_GBS,
             unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
             limiter=LIMITER.get(kernel, "HBM"), avg_launch_ms=k["avg_ms"],
             algo_bytes_per_launch=round(algo_bytes[kernel] / max(1, k["launches"])),
             fp64_vector_TFLOPs=k["algo_TFLOPs"],
             fp64_vector_frac=round(k["algo_TFLOPs"] / REAL["peak_tf"], 5))
    if r["bound"] == "vector-alu":  # priced against the vector arithmetic peak; bytes beside it
        r.update(achieved=k["algo_TFLOPs"], peak=REAL["peak_tf"], unit="TFLOP/s",
                 frac=round(k["algo_TFLOPs"] / REAL["peak_tf"], 5), hbm_GBps=achieved,
                 hbm_frac=round(achieved / HBM_PEAK_GBS, 5))
    if REAL["name"] != "f64":       # the same two fields under the names that fit
        r["vector_TFLOPs"] = r.pop("fp64_vector_TFLOPs")
        r["%s_vector_frac" % REAL["name"]] = r.pop("fp64_vector_frac")
    if peaks:
        r["peak_measured"] = peaks
        if peaks.get("hbm_triad_GBps"):
            r["frac_of_measured_hbm"] = round(k["algo_GBps"] / peaks["hbm_triad_GBps"], 5)
        if peaks.get("fp64_vector_no_fma_TFLOPs") and REAL["name"] == "f64":
            r["frac_of_measured_fp64_no_fma"] = round(
                k["algo_TFLOPs"] / peaks["fp64_vector_no_fma_TFLOPs"], 5)
    return r


# ---------------------------------------------------------------------------
# configs 2 and 4: cold solve of a batch
# ---------------------------------------------------------------------------
def run_cold(a, world, rank, local, dev, dist, peaks):
    B, N = a.instances, a.horizon
    # every rank draws its own instances: seed + rank
    d = S.make_batch(B, N, seed=a.seed + rank)
    for key in ("refs", "meas", "wind"):
        d[key] = d[key].astype(REAL["np"])
    refs = torch.from_numpy(np.ascontiguousarray(d["refs"][:, :N + 1])).to(dev)
    meas = torch.from_numpy(np.ascontiguousarray(d["meas"][:, 0])).to(dev)
    wind = torch.from_numpy(d["wind"]).to(dev)
    out_u = torch.zeros((B, 3), dtype=REAL["torch"], device=dev)
    out_st = torch.zeros(B, dtype=torch.int32, device=dev)
    out_it = torch.zeros(B, dtype=torch.int32, device=dev)

    # one explicit stream for the batch's kernels, torch's copies and the timing
    # events (torch's default stream has handle 0, which Batch() reads as "make
    # a private stream")
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    bt = Batch(B, N, device=local, stream=stream.cuda_stream)
    bt.setup(**S.setup_args())

    gather = None
    if world > 1:
        from nmpc_amd import sharding
        # results of every step go to every rank; the collective of one step runs
        # beside the kernels of the next
        gather = sharding.ResultGather([B] * world, dev)

    def step():
        bt.reset()
        bt.initialise_horizon()
        bt.set_wind_velocity(wind)
        bt.set_reference(refs)
        bt.prepare()
        bt.solve(meas)
        bt.get_controls_into(out_u, out_st, out_it)
        if gather:
            gather.start(out_u, out_st, out_it)
        return out_u, out_st, out_it

    # Every timed launch costs two event packets on the stream, so the timed region
    # times only the kernel the roofline is about: the one that took most of the
    # warm-up steps (all kernels timed there).  The per-kernel table comes from a
    # short pass after the timed region.
    bt.set_profiling(True)
    for _ in range(max(1, a.warmup)):
        step()
    torch.cuda.synchronize(dev)
    pre = {k: ms for k, (ms, n) in bt.kernel_times().items() if n and k in ALGO}
    bt.set_profiling(False)
    dominant = max(pre, key=pre.get) if pre else None
    if dist:
        dist.barrier()
    bt.set_profiling(True, kinds=[dominant] if dominant else None)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    marks[0].record(stream)
    for k in range(a.steps):
        res = step()
        marks[k + 1].record(stream)
    if gather:
        res = gather.finish()       # the last step's collective ends inside the timed region
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ktimes = bt.kernel_times()
    bt.set_profiling(False)
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps))
    table_steps = 0
    ktable = ktimes
    if dominant:
        table_steps = min(3, a.steps)
        bt.set_profiling(True)
        for _ in range(table_steps):
            step()
        if gather:
            gather.finish()
        ktable = bt.kernel_times()
        bt.set_profiling(False)
    torch.cuda.synchronize(dev)
    u_all, st_all, it_all = (x.cpu().numpy() for x in res)
    u, st, it = out_u.cpu().numpy(), out_st.cpu().numpy(), out_it.cpu().numpy()
    # NMPC_BENCH_DUMP=<file>: the gathered results of the last timed step, every
    # rank's range in rank order, for a check against the CPU oracle (tests)
    if rank == 0 and os.environ.get("NMPC_BENCH_DUMP"):
        np.savez(os.environ["NMPC_BENCH_DUMP"], u=u_all, status=st_all, iterations=it_all)

    total_instances = B * world
    value = total_instances * a.steps / elapsed

    # ---- warm control steps (reported beside the headline, outside its timed region) ----
    # The batch holds the last cold solve; every further step is the reference's
    # control loop: prepare, feedback on a new measurement, get controls, shift the
    # horizon by one reference point (which linearises one stage).
    warm = None
    if world == 1 and a.warm_steps > 0:
        dw = S.make_batch(B, N, extra_steps=2 * a.warm_steps, seed=a.seed + rank)
        wrefs = torch.from_numpy(np.ascontiguousarray(
            dw["refs"][:, N + 1:].transpose(1, 0, 2).astype(REAL["np"]))).to(dev)      # [step][B][16]
        wmeas = torch.from_numpy(np.ascontiguousarray(
            dw["meas"][:, 1:].transpose(1, 0, 2).astype(REAL["np"]))).to(dev)          # [step][B][13]
        wm = [torch.cuda.Event(enable_timing=True) for _ in range(a.warm_steps + 1)]
        bt.update_horizon(wrefs[0])
        torch.cuda.synchronize(dev)
        wm[0].record(stream)
        for k in range(a.warm_steps):
            bt.prepare()
            bt.solve(wmeas[k])
            bt.get_controls_into(out_u, out_st, out_it)
            if k + 1 < a.warm_steps:
                bt.update_horizon(wrefs[k + 1])
            wm[k + 1].record(stream)
        torch.cuda.synchronize(dev)
        wms = sorted(wm[k].elapsed_time(wm[k + 1]) for k in range(a.warm_steps))
        warm = dict(steps=a.warm_steps, p50_ms=round(percentile(wms, 0.5), 4),
                    p99_ms=round(percentile(wms, 0.99), 4),
                    steps_per_s=round(B * a.warm_steps / (sum(wms) / 1e3), 1),
                    ok_frac=round(float((out_st.cpu().numpy() == 0).mean()), 4))
        # the same loop with the real-time-iteration preparation step: every stage
        # re-linearised about the last solution before each feedback step
        bt.set_rti(True)
        bt.update_horizon(wrefs[a.warm_steps])
        torch.cuda.synchronize(dev)
        wm[0].record(stream)
        for k in range(a.warm_steps):
            bt.prepare()
            bt.solve(wmeas[a.warm_steps + k])
            bt.get_controls_into(out_u, out_st, out_it)
            if k + 1 < a.warm_steps:
                bt.update_horizon(wrefs[a.warm_steps + k + 1])
            wm[k + 1].record(stream)
        torch.cuda.synchronize(dev)
        bt.set_rti(False)
        rms = sorted(wm[k].elapsed_time(wm[k + 1]) for k in range(a.warm_steps))
        warm["rti"] = dict(p50_ms=round(percentile(rms, 0.5), 4), p99_ms=round(percentile(rms, 0.99), 4),
                           ok_frac=round(float((out_st.cpu().numpy() == 0).mean()), 4),
                           mean_qp_iterations=round(float(out_it.float().mean()), 3))

    # ---- roofline of the dominant kernel (rank 0's GPU) ----------------------
    # units each launch kind processed over the timed steps
    newton_runs = np.where(st == 0, it - 1, it).astype(np.int64)    # factor/step executions
    setup_runs = it.astype(np.int64)                                # gradient evaluations

    # algorithmic bytes / flop each kernel kind processed over the timed steps; the Newton
    # set-up runs fused behind qp_begin (iteration 1) and newton_step (later iterations)
    def rates(times, steps):
        """Per-kernel totals and algorithmic rates of `steps` steps."""
        n_setup_first = B * N * steps
        n_setup_later = (int(setup_runs.sum()) - B) * N * steps
        n_step = int(newton_runs.sum()) * (N + 1) * steps
        work = {
            "linearise": [(ALGO["linearise"], B * N * steps)],
            "qp_begin": [(ALGO["qp_begin"], B * (N + 1) * steps), (ALGO["newton_setup"], n_setup_first)],
            "newton_solve": [(ALGO["newton_solve"], int(newton_runs.sum()) * N * steps),
                             (ALGO["newton_factor"], int((newton_runs >= 1).sum()) * N * steps)],
            "newton_step": [(ALGO["newton_step"], n_step), (ALGO["newton_setup"], n_setup_later)],
        }
        kern, algo_bytes = {}, {}
        for name, (ms, n) in times.items():
            if n == 0:
                continue
            if name not in work:
                kern[name] = dict(total_ms=round(ms, 3), launches=int(n), avg_ms=round(ms / n, 4))
                continue
            nbytes = sum(al["bytes"] * u for al, u in work[name]) * REAL["bytes"] / 8.0
            nflop = sum(al["flop"] * u for al, u in work[name])
            algo_bytes[name] = nbytes
            kern[name] = dict(total_ms=round(ms, 3), launches=int(n), avg_ms=round(ms / n, 4),
                              algo_GBps=round(nbytes / 1e9 / (ms / 1e3), 1),
                              algo_TFLOPs=round(nflop / 1e12 / (ms / 1e3), 3))
        return kern, algo_bytes

    # the roofline kernel over the timed region; every kernel over the pass after it
    timed_kern, algo_bytes = rates(ktimes, a.steps)
    if dominant is None:
        dominant = max(algo_bytes, key=lambda k: timed_kern[k]["total_ms"])
    kern = rates(ktable, table_steps)[0] if table_steps else timed_kern
    kern[dominant] = timed_kern[dominant]
    tag = "cfg%d" % a.config if (B, N) in ((4096, 100), (1024, 500)) else "b%d_n%d" % (B, N)
    roofline = make_roofline(dominant, kern, algo_bytes, peaks, load_traffic(tag, dominant))
    whole = whole_step_traffic(tag, elapsed / a.steps * 1e3, peaks)
    if whole:
        roofline["whole_step"] = whole
    # the stage linearisation beside it, whichever kernel dominates
    if "linearise" in kern and dominant != "linearise":
        lin_bytes = {"linearise": ALGO["linearise"]["bytes"] * REAL["bytes"] / 8.0 * B * N
                                  * kern["linearise"]["launches"]}
        roofline["linearise"] = make_roofline("linearise", kern, lin_bytes, None,
                                              load_traffic(tag, "linearise"))

    # ---- CPU baseline (rank 0, N=1 only) -------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import portdriver as P
        cores, host = host_cores(a.cpu_threads)
        n = min(a.cpu_sample or 4096, B)
        cfg = S.setup_args()
        # repeat the sample until about 1.5 s of wall time (~20 core-seconds) has been timed
        secs, passes = 0.0, 0
        while secs < 1.5 and passes < 64:
            cu, cst, cit, s1 = P.batch_cold_solve(
                N, cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                cfg["lower_control_bound"], cfg["upper_control_bound"], d["refs"][:n, :N + 1],
                d["meas"][:n, 0], d["wind"][:n], threads=cores, precision=REAL["precision"])
            secs += s1
            passes += 1
        parity = bool(same_bits(cu, u[:n]) and np.array_equal(cst, st[:n])
                      and np.array_equal(cit, it[:n]))
        cpu = dict(value=round(n * passes / secs, 1), unit="solves/s", cores=cores, kind="port",
                   sample="first %d instances of the same workload, cold solve each, %d passes, "
                          "%.2f s wall on %d threads" % (n, passes, secs, cores),
                   gpu_results_identical=parity, **host)

    line = None
    if rank == 0:
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "solves/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "p50_step_ms": round(percentile(step_ms, 0.5), 4),
            "p99_step_ms": round(percentile(step_ms, 0.99), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": REAL["name"], "data": "synthetic",
            "config": {"workload": "%d independent UAV NMPC instances per GPU, horizon %d, "
                                   "cold solve (reset, init, %d reference points = all stages "
                                   "linearised, feedback, get controls), %s"
                                   % (B, N, N + 1, REAL["name"]),
                       "baseline_config": a.config,
                       "instances_per_gpu": B, "horizon": N, "total_instances": total_instances,
                       "parallelism": "instance partition, %d GPU(s)" % world},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "warm_step": warm,
            "kernels": kern,
            "kernels_from": ("%s: the %d timed steps; the others: %d further steps with every "
                             "kernel timed" % (dominant, a.steps, table_steps)) if table_steps
                            else "the timed steps, every kernel timed",
            "solver": {"converged_frac": round(float((st_all == 0).mean()), 4),
                       "mean_qp_iterations": round(float(it_all.mean()), 3)},
        }
    bt.close()
    return line


# ---------------------------------------------------------------------------
# config 5: stage linearisation of independent samples
# ---------------------------------------------------------------------------
def run_samples(a, world, rank, local, dev, dist, peaks):
    n = a.samples
    rng = np.random.default_rng(a.seed + rank)
    x, u, w = (v.astype(REAL["np"]) for v in S.random_states(n, rng))
    xn = np.roll(x, 1, axis=0)
    lo, up = np.zeros(3, REAL["np"]), np.ones(3, REAL["np"])
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    tx, tu, tn, tw = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x, u, xn, w))
    jac = torch.empty((n, 180), dtype=REAL["torch"], device=dev)
    res = torch.empty((n, 12), dtype=REAL["torch"], device=dev)
    lib = _lib.load()
    plo, pup = lo.ctypes.data_as(lib.preal), up.ctypes.data_as(lib.preal)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    hs = C.c_void_p(stream.cuda_stream)

    def step():
        _lib.check(lib, lib.nmpc_linearise_samples(local, hs, n, ptr(tx), ptr(tu), ptr(tn), ptr(tw),
                                                   plo, pup, ptr(jac), ptr(res)))

    for _ in range(max(1, a.warmup)):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record(stream)
    for k in range(a.steps):
        step()
        marks[k + 1].record(stream)     # each step is one launch: these are launch durations
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps))
    avg_ms = sum(step_ms) / len(step_ms)
    al = dict(flop=ALGO["linearise"]["flop"], bytes=ALGO["linearise"]["bytes"] * REAL["bytes"] / 8.0)
    kern = {"linearise": dict(total_ms=round(sum(step_ms), 3), launches=a.steps,
                              avg_ms=round(avg_ms, 4),
                              algo_GBps=round(al["bytes"] * n / 1e9 / (avg_ms / 1e3), 1),
                              algo_TFLOPs=round(al["flop"] * n / 1e12 / (avg_ms / 1e3), 3))}
    roofline = make_roofline("linearise", kern, {"linearise": al["bytes"] * n * a.steps}, peaks,
                             load_traffic("cfg5", "linearise_free"))

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import portdriver as P
        cores, host = host_cores(a.cpu_threads)
        m = min(a.cpu_sample or (1 << 18), n)
        secs, passes = 0.0, 0
        while secs < 1.5 and passes < 16:
            cj, cr, s1 = P.batch_linearise(x[:m], u[:m], xn[:m], w[:m], lo, up, threads=cores,
                                           precision=REAL["precision"])
            secs += s1
            passes += 1
        gj, gr = jac[:m].cpu().numpy(), res[:m].cpu().numpy()
        cpu = dict(value=round(m * passes / secs, 1), unit="stage linearisations/s", cores=cores,
                   kind="port",
                   sample="first %d samples of the same workload, %d passes, %.2f s wall on %d "
                          "threads" % (m, passes, secs, cores),
                   gpu_results_identical=bool(same_bits(cj, gj) and same_bits(cr, gr)), **host)
    if rank != 0:
        return None
    return {
        "metric": "stage linearisations/sec (RK4 + finite-difference Jacobian per sample)",
        "value": round(n * world * a.steps / elapsed, 1), "unit": "stage linearisations/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "p50_step_ms": round(percentile(step_ms, 0.5), 4),
        "p99_step_ms": round(percentile(step_ms, 0.99), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": REAL["name"],
        "data": "synthetic",
        "config": {"workload": "%d independent (x, u, wind) samples per GPU through the stage "
                               "linearisation (1 nominal + 15 perturbed RK4 integrations, "
                               "12x15 Jacobian and residual out), %s" % (n, REAL["name"]),
                   "baseline_config": 5, "samples_per_gpu": n,
                   "parallelism": "sample partition, %d GPU(s)" % world},
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kern,
    }


# ---------------------------------------------------------------------------
# config 1: one controller through the reference's own interface
# ---------------------------------------------------------------------------
def run_single(a, local, dev, peaks):
    os.environ["NMPC_DEVICE"] = str(local)
    import nmpc_amd as nmpc
    N = 100
    steps = a.steps
    d = S.make_batch(1, N, extra_steps=steps + a.warmup + 1, seed=a.seed)
    refs, meas, wind = d["refs"][0], d["meas"][0], d["wind"][0]
    cfg = S.setup_args()
    nmpc.init()
    nmpc.setup(**{k: list(v) for k, v in cfg.items()})
    t0 = time.perf_counter()
    nmpc.initialise_horizon()
    nmpc.set_wind_velocity(list(wind))
    for i in range(N + 1):
        nmpc.set_reference(list(refs[i]), i)
    nmpc.prepare()
    nmpc.solve(list(meas[0]))
    u0 = nmpc.get_controls()
    cold_ms = (time.perf_counter() - t0) * 1e3

    def control_step(k):
        nmpc.update_horizon(list(refs[N + 1 + k]))
        nmpc.prepare()
        nmpc.set_wind_velocity(list(wind))
        nmpc.solve(list(meas[k + 1]))
        return nmpc.get_result()

    for k in range(a.warmup):
        control_step(k)
    lat, us, sts = [], [], []
    tstart = time.perf_counter()
    for k in range(a.warmup, a.warmup + steps):
        t1 = time.perf_counter()
        uk, stk = control_step(k)
        lat.append((time.perf_counter() - t1) * 1e3)
        us.append(uk)
        sts.append(stk)
    elapsed = time.perf_counter() - tstart
    lat.sort()

    # the same loop on one host core: the compiled reference where it travelled
    # with the tree, otherwise the oracle port
    cpu = None
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        kind = "port"
        ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_n100.so")
        if os.path.exists(ref_so):
            import refdriver
            drv = refdriver.Ref(100)
            kind = "reference"
        else:
            import portdriver
            drv = portdriver.Port(100)
        drv.setup(cfg["state_weights"], cfg["control_weights"], cfg["terminal_weights"],
                  cfg["upper_control_bound"], cfg["lower_control_bound"])
        drv.init()
        drv.set_wind(wind)
        for i in range(N + 1):
            drv.set_reference(refs[i], i)
        drv.feedback(meas[0])
        cu0, _ = drv.controls()
        clat, same = [], same_bits(np.asarray(cu0, np.float64), np.asarray(u0, np.float64))
        for k in range(a.warmup + steps):
            t1 = time.perf_counter()
            drv.update_horizon(refs[N + 1 + k])
            drv.set_wind(wind)
            drv.feedback(meas[k + 1])
            cu, cst = drv.controls()
            if k >= a.warmup:
                clat.append((time.perf_counter() - t1) * 1e3)
                j = k - a.warmup
                same = same and bool(same_bits(np.asarray(cu, np.float64), np.asarray(us[j], np.float64))
                                     and cst == sts[j])
        clat.sort()
        cpu = dict(value=round(1e3 * len(clat) / sum(clat), 1), unit="solves/s", cores=1, kind=kind,
                   p50_ms=round(percentile(clat, 0.5), 4), p99_ms=round(percentile(clat, 0.99), 4),
                   sample="the same %d control steps of the same controller, one thread, "
                          "through ctypes" % len(clat),
                   gpu_results_identical=same, **host_cores(1)[1])
    return {
        "metric": METRIC, "value": round(steps / elapsed, 1), "unit": "solves/s", "n_gpus": 1,
        "steps": steps, "warmup": a.warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
        "p50_step_ms": round(percentile(lat, 0.5), 4), "p99_step_ms": round(percentile(lat, 0.99), 4),
        "cold_start_ms": round(cold_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "one UAV NMPC controller, horizon 100, through the reference's "
                               "24-symbol C interface (Python module mirror): control step = "
                               "update_horizon, prepare, set wind, feedback, get controls; host "
                               "wall time per step, fp64",
                   "baseline_config": 1, "instances_per_gpu": 1, "horizon": N},
        "roofline": dict(kernel="whole control step", bound="launch-latency", achieved=None,
                         peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None,
                         note="one instance keeps one workgroup busy; the step is a fixed "
                              "sequence of dependent launches, not a bandwidth or flop question",
                         peak_measured=peaks),
        "cpu_baseline": cpu,
        "solver": {"converged_frac": round(float(np.mean([s == 0 for s in sts])), 4)},
    }


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                     % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    REAL["torch"] = torch.float64
    if a.precision == "float":
        if a.config == 1:
            sys.exit("--precision float: configs 2, 4 and 5 only")
        assert _lib.load().np_real == "float32", "libcnmpc_f32.so was not the library loaded"
        REAL.update(name="f32", np=np.float32, torch=torch.float32, bytes=4, precision="float",
                    peak_tf=FP32_VECTOR_PEAK_TF)
    # Rehearsal of the multi-rank path on a box with fewer GPUs than ranks:
    # NMPC_BENCH_REHEARSE=1 lets ranks share GPUs and gathers over gloo (RCCL
    # refuses two ranks on one device).  Never set for a measurement.
    rehearse = os.environ.get("NMPC_BENCH_REHEARSE") == "1"
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    peaks = None
    if rank == 0 and not a.no_peaks:
        peaks = measure_peaks(local)
    if dist:
        dist.barrier()

    if a.config == 1:
        if world > 1:
            sys.exit("config 1 is a single controller: run it with --gpus 1")
        line = run_single(a, local, dev, peaks)
    elif a.config == 5:
        line = run_samples(a, world, rank, local, dev, dist, peaks)
    else:
        line = run_cold(a, world, rank, local, dev, dist, peaks)
    if rank == 0:
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
