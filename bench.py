#!/usr/bin/env python3
"""bench.py -- self-play positions/sec of the MI355X-native hot path (BASELINE.json metric).

Workload (BASELINE.json configs[1]): GoEnv(9), tower_height=10, 400 readouts, 1024 concurrent
self-play games per GPU, synthetic Flux-default-equivalent weights (random init from the draw
stream), exact-f32 MFMA network.  One "step" = one tree_search! round for every live game: select
up to 8 leaves per game -> 17-plane features -> ResNet forward on the coalesced batch (<= 8192
positions) -> expand / back up, plus the per-move phase (resign check, pick, play, re-root, noise)
for the games whose 400-readout budget is spent.  A "position" = one self-play move that received
its full 400 readouts inside the run.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  value = positions played by all ranks / max-over-ranks time of
exactly K steps bracketed by barrier + device sync.  Extra objects:
  roofline      the dominant kernel (3x3 256->256 conv, v_mfma_f32_32x32x2_f32): algorithmic
                TFLOP/s from HIP events around every tower-conv launch of the timed region
  cpu_baseline  the CPU oracle (a port of the reference algorithm in the reference's execution
                shape: one game at a time, batches of 8) timed on this host, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md chip table (f32-in MFMA = f32 vector peak)


def f_eval(N, t):
    """algorithmic FLOP per network evaluation"""
    P = N * N
    return 2.0 * P * (9 * 17 * 256 + t * 2 * 9 * 256 * 256) + 2.0 * P * 256 * 3 + 2.0 * (2 * P * (P + 1) + P * 256 + 256)


def pmc_traffic(rows_per_launch, N, precision="f32", f43=False):
    """HBM bytes per tower-conv launch, from the committed rocprofv3 --pmc passes.

    Hardware counters cannot be read from inside this process; FETCH_SIZE and WRITE_SIZE were collected in two
    separate `rocprofv3 --pmc` runs of THIS command (tools/profile_r06.sh: same kernels, same batch) and summarised
    by tools/pmc_traffic.py as bytes per board-point row.  Scaled here by the average rows per launch of THIS run.
    The newest round's file for the board size and precision wins; the F(4x4,3x3) tower (19x19 f32 since round 4) only
    takes a round-4 or later file.  None if no file exists."""
    names = [f"r06_pmc_traffic_{N}x{N}_{precision}.json", f"r05_pmc_traffic_{N}x{N}_{precision}.json",
             f"r04_pmc_traffic_{N}x{N}_{precision}.json"]
    if not f43:
        names.append(f"r03_pmc_traffic_{N}x{N}_{precision}.json")
        if N == 9:
            names.append("pmc_traffic.json" if precision == "f32" else f"r02_pmc_traffic_{precision}.json")
    if N >= 13 and precision == "f32" and not f43:
        names = names[3:]              # (--winograd 2 on a large board: the round-3 kernel's file)
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
            return d["bytes_per_row"] * rows_per_launch, f"profiles/{name} ({d['source']})"
        except Exception:
            continue
    return None, None


def live_pmc_traffic(args, seconds_cap=240.0):
    """HBM bytes per tower layer measured NOW, on this box: two child runs of this very command under `rocprofv3 --pmc`
    (FETCH_SIZE, then WRITE_SIZE: separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes), a few steps each
    with one tower chain so that a dispatch is a layer, summarised by tools/pmc_traffic.py (FETCH doubled for 16 B/lane
    streaming reads).  Returns (bytes_per_row, source) or (None, why).  Hardware counters cannot be read from inside a
    process that is not being profiled, hence the children; the parent's engine is idle meanwhile."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "no rocprofv3 on this host"
    spec = importlib.util.spec_from_file_location("agz_pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
    pt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pt)
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix="agz_pmc_", dir="/tmp") as tmp:
        dirs = []
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--board", str(args.board), "--tower", str(args.tower),
                   "--readouts", str(args.readouts), "--games", str(args.games), "--stagger", str(args.stagger),
                   "--precision", args.precision, "--winograd", str(args.winograd), "--tower-streams", "1", "--no-cpu-baseline",
                   "--no-alt-precision", "--no-config-legs", "--generation", "0", "--no-sustained", "--no-live-traffic"]
            left = seconds_cap - (time.time() - t0)
            if left < 20:
                return None, f"counter passes over their {seconds_cap:.0f} s budget"
            try:
                r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} pass timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr[-200:]}"
            dirs.append(d)
        try:
            d = pt.summarise(8 * args.games, args.board, dirs)
        except Exception as ex:
            return None, f"{type(ex).__name__}: {ex}"
    live_pmc_traffic.search = d.get("search_kernels") or None      # (per-dispatch bytes of k_pre ... k_post of the same passes)
    return d["bytes_per_row"], (f"live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this command on this box, "
                                f"{d['tower_layer_dispatches']} tower-layer dispatches, {time.time() - t0:.0f} s")


live_pmc_traffic.search = None


def search_kernels_object(search_ms, search_steps, d, games, N, step_ms, pmc):
    """the search / feature / legal kernels as microseconds per step and HBM GB/s (these are integer / byte
    paths: the roofline that applies is HBM's ~8 TB/s, and for the tree walks of k_pre / k_post not even that but the latency of
    a chain of dependent loads).  Time: HIP events on the engine's stream around each kernel of the timed steps.  Bytes:
    `algorithmic` where the kernel has a closed form -- k_leaf_features reads 8 boards of N^2 int8 and writes N^2 x 32 f32 per
    leaf (features.jl:3-26), k_expand reads and writes one board, one legal mask and one 32-byte node record per new leaf
    (board.jl:451-509), k_scan one count and one base per game -- and `pmc` = FETCH_SIZE + WRITE_SIZE per dispatch from the
    live rocprofv3 counter passes of this command (as counted: no streaming-read correction applies to 1-8 B/lane loads)."""
    if not search_steps:
        return None
    P, A = N * N, N * N + 1
    leaves = d["evals"] / max(d["steps"], 1)                     # network rows per step = leaves collected per step
    alg = {"k_leaf_features": leaves * (8.0 * P + 128.0 * P),
           "k_expand": leaves * (2.0 * P + 4.0 * ((A + 31) // 32) + 64.0),
           "k_scan": games * 8.0}
    out = {"steps_timed": search_steps, "leaves_per_step": leaves, "hbm_peak_gb_s": 8000.0, "kernels": {}}
    tot = 0.0
    for k, ms in search_ms.items():
        us = 1e3 * ms / search_steps
        tot += us
        o = {"us_per_step": us, "share_of_step": us * 1e-3 / step_ms}
        if k in alg:
            o["algorithmic_bytes_per_step"] = alg[k]
            o["algorithmic_gb_s"] = alg[k] / (us * 1e-6) / 1e9 if us > 0 else None
            o["algorithmic_frac_of_hbm_peak"] = o["algorithmic_gb_s"] / 8000.0 if us > 0 else None
        if pmc and k in pmc:
            b = sum(v for c, v in pmc[k].items() if c.endswith("_bytes_per_dispatch"))
            o["pmc_bytes_per_dispatch"] = b
            o["pmc_gb_s"] = b / (us * 1e-6) / 1e9 if us > 0 else None
            o["pmc_frac_of_hbm_peak"] = o["pmc_gb_s"] / 8000.0 if us > 0 else None
        out["kernels"][k] = o
    out["us_per_step_total"] = tot
    out["share_of_step_total"] = tot * 1e-3 / step_ms
    out["bound"] = ("k_pre / k_post: latency of dependent loads down one tree per wave (one round trip per tree level); "
                    "k_expand: LDS latency of two component labellings per new leaf; k_leaf_features: HBM writes")
    out["pmc_source"] = ("live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (the passes behind roofline.traffic)"
                         if pmc else None)
    return out


class PowerSampler:
    """Socket power and shader clock of one GPU, sampled from the amdgpu hwmon files while a timed region runs.

    The f32 MFMA layer is power-limited (HISTORY.md 4f): its `frac` of the nominal peak only means something next to the
    clock and the watts the board actually ran at, so the line carries them.  Sources, first that works:
    hwmon power1_average / power1_input (microwatts) and freq1_input (Hz, sclk) under the device's sysfs node; the
    amdsmi Python module; nothing (the object then says so).  ~20 samples per second on a daemon thread: no GPU work."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self.samples, self._stop, self._th = [], threading.Event(), None
        self.source, self._pw, self._fq, self._smi = None, None, None, None
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if hw:
                cards.append((os.path.realpath(dev), hw[0]))
        want = None
        try:      # match the torch device by PCI address where the build exposes it
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:
            pass
        pick = [c for c in cards if want and want in c[0]] or cards[device_index:device_index + 1] or cards[:1]
        if pick:
            hw = pick[0][1]
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(hw, name)):
                    self._pw = os.path.join(hw, name)
                    break
            if os.path.exists(os.path.join(hw, "freq1_input")):
                self._fq = os.path.join(hw, "freq1_input")
            if self._pw or self._fq:
                self.source = f"sysfs {hw} ({os.path.basename(self._pw) if self._pw else '-'}, {'freq1_input' if self._fq else '-'})"
        if self.source is None:
            try:
                import amdsmi
                amdsmi.amdsmi_init()
                self._smi = (amdsmi, amdsmi.amdsmi_get_processor_handles()[device_index])
                self.source = "amdsmi"
            except Exception:
                self._smi = None

    def _read(self):
        w = mhz = None
        try:
            if self._pw:
                w = int(open(self._pw).read()) * 1e-6
            if self._fq:
                mhz = int(open(self._fq).read()) * 1e-6
            if self._smi:
                m, h = self._smi
                pw = m.amdsmi_get_power_info(h)
                w = float(pw.get("current_socket_power") or pw.get("average_socket_power") or 0) or None
                ck = m.amdsmi_get_clock_info(h, m.AmdSmiClkType.GFX)
                mhz = float(ck.get("clk") or ck.get("cur_clk") or 0) or None
        except Exception:
            pass
        return w, mhz

    def start(self):
        import threading
        if self.source is None:
            return self
        self.samples = []
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._read())
                self._stop.wait(0.05)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def stop(self):
        if self._th is not None:
            self._stop.set()
            self._th.join(timeout=2.0)
            self._th = None
        ws = [w for w, _ in self.samples if w]
        fs = [f for _, f in self.samples if f]
        if self.source is None:
            return {"source": None, "note": "no hwmon power/clock file and no amdsmi module on this host"}
        return {"source": self.source, "samples": len(self.samples),
                "socket_power_w": {"mean": sum(ws) / len(ws), "max": max(ws)} if ws else None,
                "sclk_mhz": {"mean": sum(fs) / len(fs), "min": min(fs), "max": max(fs)} if fs else None}


def cpu_baseline(N, tower, readouts, seconds):
    """oracle selfplay on the host cores; returns the cpu_baseline object"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import numpy as np
    import orc

    L = orc.lib()
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:  # a cgroup CPU quota is invisible to affinity: honour it
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = max(1, min(avail, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    net = L.or_net_new(N, tower)
    L.or_net_init_synthetic(net, 0)
    A = N * N + 1
    x = np.zeros((8, 17 * N * N), np.float32)
    pi = np.zeros((8, A), np.float32)
    v = np.zeros(8, np.float32)
    # pick the thread count that makes one batch-of-8 forward fastest on THIS host
    best = (None, 1e30)
    for cores in sorted({avail, max(1, avail // 2), 64, 32, 16, 8}, reverse=True):
        if cores > avail:
            continue
        L.or_set_num_threads(cores)
        L.or_net_forward_feats(net, orc.fptr(x), 8, orc.fptr(pi), orc.fptr(v), 32)
        t0 = time.perf_counter()
        L.or_net_forward_feats(net, orc.fptr(x), 8, orc.fptr(pi), orc.fptr(v), 32)
        dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (cores, dt)
        if dt > 4 * best[1] and dt > 0.5:
            break
    cores, t8 = best
    L.or_set_num_threads(cores)
    per_move = t8 * (readouts / 8.0 + 1.0)
    moves = int(max(1, min(64, round(seconds / max(per_move, 1e-6)))))
    cb = orc.NET_FN(lambda ctx, pos, B, ppi, pv: L.or_net_callable(net, pos, B, ppi, pv))
    t0 = time.perf_counter()
    p = L.or_selfplay(N, cb, None, readouts, 1, 0, moves)
    dt = time.perf_counter() - t0
    played = L.or_player_num_moves(p)
    evals = L.or_player_evals(p)
    L.or_player_free(p)
    L.or_net_free(net)
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {
        "value": played / dt, "unit": "positions/s", "cores": cores, "kind": "port", "cpu": model,
        "host_threads_visible": os.cpu_count(), "threads_usable": avail,
        "sample": f"first {played} moves of one self-play game (GoEnv({N}), tower {tower}, {readouts} readouts, "
                  f"batches of 8, fp32 C network with OpenMP on {cores} threads), {evals} evals in {dt:.1f} s",
        "note": "reference (Julia/Flux) is not runnable here: no julia binary; CPU restatement timed",
    }


def shard_leg(ag, torch, name, N, tower, R, games, precision, steps, warmup, stagger, device):
    """One bounded leg of another BASELINE config on this GPU, reported beside `value`, never as it: one GPU's shard of
    configs[3] (GoEnv(19), tower 20, 800 readouts, 2048 games over 8 GPUs -> 256 per GPU, exact f32) or configs[4] (fp16
    MFMA tower, 1600 readouts, 4096 games -> 512 per GPU).  Same protocol as the headline: untimed prelude to the steady
    regime, W warm-up steps, K timed steps between device synchronisations; HIP events around every tower layer; socket
    power and shader clock sampled over the K steps."""
    eng = ag.Engine(board_size=N, tower_height=tower, games=games, num_readouts=R, parallel_readouts=8, seed=1,
                    device=device, stagger_moves=stagger)
    try:
        eng.init_synthetic(0)
        eng.set_precision(precision)
        eng.start(0)
        prelude = (R + 7) // 8 + 5 if stagger > 0 else 0
        eng.step(prelude + warmup)
        eng.sync()
        s0 = eng.stats()
        eng.profile_conv(True)
        sampler = PowerSampler(device).start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(steps)
        eng.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        power = sampler.stop()
        conv_ms, conv_flop, conv_n = eng.profile_conv_read()
        eng.profile_conv(False)
        s1 = eng.stats()
        f16 = precision == "f16"
        T4 = (N + 3) // 4
        ratio = 1.0 if f16 else 36.0 * T4 * T4 / (9.0 * N * N)          # F(4x4,3x3): 36 multiplies per 4x4 tile (N >= 13)
        peak = 2500.0 if f16 else PEAK_F32_MFMA_TFLOPS
        alg = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None
        pos = s1["positions"] - s0["positions"]
        try:        # what this board sustains on MFMAs alone with changing operands, in this leg's arithmetic
            sustained = eng.mfma_sustained_data_tflops(400, 2 if f16 else 3)      # fp16: dense weights x half-zero activations
        except Exception:
            sustained = None
        traffic, traffic_src = pmc_traffic(conv_flop / max(conv_n, 1) / (2.0 * 9 * 256 * 256), N, precision, not f16)
        return {
            "config": name, "precision": precision,
            "workload": f"GoEnv({N}), tower_height={tower}, {R} readouts, {games} concurrent games on this GPU "
                        f"(batch <= {8 * games} positions)",
            "value": pos / dt, "unit": "positions/s", "steps": steps, "warmup": warmup, "setup_prelude_steps": prelude,
            "ms_per_step": 1e3 * dt / steps, "positions": pos, "evals": s1["evals"] - s0["evals"],
            "layer_ms": conv_ms / max(conv_n, 1), "layers_timed": conv_n,
            "roofline": {"bound": "mfma", "kernel": "k_conv3x3_f16_w2 (implicit GEMM, v_mfma_f32_32x32x16_f16)" if f16 else
                                  "Winograd F(4x4,3x3) tower layer (v_mfma_f32_32x32x2_f32; every kernel of a layer timed together)",
                         "achieved": alg * ratio if alg else None, "peak": peak, "unit": "TFLOP/s",
                         "frac": alg * ratio / peak if alg else None, "achieved_algorithmic": alg,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": (1280.0 if f16 else 2560.0) * conv_flop / max(conv_n, 1) / (2.0 * 9 * 256 * 256),
                         "sustained_mfma_changing_operands": None if not (sustained and alg) else {
                             "value": sustained, "unit": "TFLOP/s", "frac_of_nominal_peak": sustained / peak,
                             "achieved_over_sustained": alg * ratio / sustained}},
            "end_to_end_algorithmic_tflops": pos / dt * R * f_eval(N, tower) / 1e12,
            "power": power,
            "pool": {"node_capacity": s1["node_capacity"], "peak_nodes_per_game": s1["peak_nodes_per_game"],
                     "short_searches": s1["pool_short_searches"], "refused_allocations": s1["pool_exhausted"]},
        }
    finally:
        eng.close()


def self_launch(n, single_device):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here -- N copies of
    this very command, one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their
    environment exactly as torch.distributed.run would set them -- and wait.  Rank 0 inherits this process's stdout
    (the one JSON line); the other ranks' stdout goes to stderr.  The games are independent
    (/root/reference/src/train.jl:56-57): nothing but the rendezvous address is shared.  Returns the exit status."""
    import socket
    import subprocess
    if not single_device:
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception:
            have = 0
        if have < n:
            print(f"bench.py: --gpus {n} but {have} GPU(s) visible (one rank per GPU; --single-device-test puts every "
                  f"rank on cuda:0 for a rehearsal)", file=sys.stderr)
            return 2
    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    # The port is picked by bind-then-close: another process can take it before rank 0's store binds it.  A rank
    # whose rendezvous fails exits with 75 (EX_TEMPFAIL); only then are the ranks started again, on a new port, up to three times.
    rc = 1
    for attempt in range(3):
        port = free_port()
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                       AGZ_BENCH_SELF_LAUNCHED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))
        deadline = None
        while any(p.poll() is None for p in procs):
            for p in procs:
                if p.poll() not in (None, 0) and deadline is None:
                    deadline = time.time() + 30.0                      # one rank died: the others get 30 s to follow
            if deadline is not None and time.time() > deadline:
                for p in procs:                                        # (exactly the processes started above)
                    if p.poll() is None:
                        p.kill()
            time.sleep(0.2)
        codes = [p.returncode for p in procs]
        if all(c == 0 for c in codes):
            return 0
        # a child killed by a signal reports a negative code: the launcher's own status is 1 for any failure
        rc = 1
        if 75 not in codes:
            break                                                      # not a rendezvous failure: do not run the bench twice
        print(f"bench.py: rendezvous failed (ranks exited with {codes}, port {port}); starting them again on a new port", file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--tower", type=int, default=10)
    ap.add_argument("--readouts", type=int, default=400)
    ap.add_argument("--games", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--stagger", type=int, default=60, help="random opening prefix (moves) so games are at mixed stages")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "f32s"],
                    help="tower arithmetic: f32 (default, the BASELINE metric), f16 = the fp16 MFMA path of BASELINE configs[4], "
                         "f32s = the f32 network with Winograd operands split into two f16 halves (fp16 MFMA, f32-grade results)")
    ap.add_argument("--winograd", type=int, default=1, choices=[0, 1, 2, 3],
                    help="agz_net_set_winograd: 1 = default (F(4x4,3x3) from 13x13 up in exact f32, else F(3x3,3x3)), "
                         "2 = F(3x3,3x3) everywhere, 0 = direct implicit GEMM")
    ap.add_argument("--tower-streams", type=int, default=2, choices=[1, 2, 3, 4],
                    help="agz_net_set_tower_streams: the F(4x4,3x3) tower as 2 (default) or 1 layer chains")
    ap.add_argument("--tower-persistent", action="store_true",
                    help="run the f32 Winograd tower as one persistent launch (agz_net_set_tower_persistent; same bits, HISTORY.md 4f)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the 0.4 s sustained-MFMA-rate measurement behind the timed region")
    ap.add_argument("--no-alt-precision", action="store_true",
                    help="skip the extra (untimed for `value`) leg that repeats the K steps with --precision f32s")
    ap.add_argument("--generation", type=int, default=None, metavar="G",
                    help="after the timed K steps keep playing, untimed for `value`: a warm-up until G games have ended "
                         "naturally, then a window until G more have; the line gains a `generation` object with SURVEY.md 8d's "
                         "generation rate (sum of position.n of the games that ended in the window / wall time) next to the "
                         "steady-state rate of the same window.  Default: a whole generation (G = --games = 1024) on the headline "
                         "workload at N = 1 (0 = off), bounded by --generation-seconds")
    ap.add_argument("--generation-seconds", type=float, default=900.0,
                    help="hard cap on the generation leg (half for the warm-up generation, half for the measured one); a leg the "
                         "cap cut short reports what it saw and says so")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the committed profiles/ file instead of two live rocprofv3 --pmc passes of this "
                         "command (the default at N = 1 on the headline workload when rocprofv3 is installed)")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the two bounded legs that run one GPU's shard of BASELINE configs[3] (19x19, tower 20, 800 "
                         "readouts, 256 games, f32) and configs[4] (fp16 tower, 1600 readouts, 512 games) after the headline")
    ap.add_argument("--config-leg-steps", type=int, default=20)
    ap.add_argument("--single-device-test", action="store_true",
                    help="testing only: every rank uses cuda:0 and gloo, to exercise the multi-rank code path on a 1-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher: be the launcher
        sys.exit(self_launch(args.gpus, args.single_device_test))

    # stdout carries exactly ONE JSON line: gloo / RCCL / the HIP runtime print banners on fd 1 from native code, so fd 1
    # points at stderr for the whole run and the line goes out through a private copy of the original stdout
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    import alphago_jl_amd as ag

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    headline = (args.board, args.tower, args.readouts, args.games, args.precision) == (9, 10, 400, 1024, "f32")
    if args.generation is None:      # SURVEY.md 8d's definition rides along on the headline line
        args.generation = args.games if (world == 1 and headline and args.stagger > 0) else 0
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher that set WORLD_SIZE={world}: start {args.gpus} ranks, or "
                         f"run `python bench.py --gpus {args.gpus}` with WORLD_SIZE unset and it starts them itself")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.single_device_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    rdev = "cuda"          # where the cross-rank reductions of the two scalars live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            if args.single_device_test:
                dist.init_process_group("gloo")
                rdev = "cpu"
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        except Exception as ex:
            if os.environ.get("AGZ_BENCH_SELF_LAUNCHED") == "1":
                # the self-launcher picked MASTER_PORT by bind-then-close: if somebody took it meanwhile, say so with a status
                # the launcher recognises (EX_TEMPFAIL) and it starts the ranks again on another port
                print(f"bench.py: rendezvous failed on rank {rank}: {type(ex).__name__}: {ex}", file=sys.stderr)
                sys.exit(75)
            raise

    N, tower, R = args.board, args.tower, args.readouts
    eng = ag.Engine(board_size=N, tower_height=tower, games=args.games, num_readouts=R, parallel_readouts=8,
                    seed=1, game_id_base=rank, game_id_stride=world, device=local_rank,
                    stagger_moves=args.stagger, record_capacity_games=max(2 * args.games + 64, 2 * (args.generation or 0) + 64))
    eng.init_synthetic(0)
    eng.set_precision(args.precision)
    if args.winograd != 1:
        eng.set_winograd(args.winograd)
    if args.tower_streams != 2:
        eng.set_tower_streams(args.tower_streams)
    if args.tower_persistent:
        eng.set_tower_persistent(True)
    eng.start(0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Setup, untimed and separate from the W warm-up steps: bring the synthetic state to the
    # steady regime.  With --stagger every initial game gets a random opening prefix and a random
    # fraction of its first readout budget (that shortened first move is never counted as a
    # position), so after ceil(R/8) steps every game is mid-search at a uniformly random phase and
    # K timed steps see K*8/R completed full-budget moves per game on average, for any K.
    prelude = (R + 7) // 8 + 5 if args.stagger > 0 else 0
    if prelude:
        eng.step(prelude)
    eng.step(args.warmup)
    eng.sync()
    s0 = eng.stats()
    eng.profile_conv(True)
    eng.profile_search(True)
    sampler = PowerSampler(local_rank) if rank == 0 else None
    barrier()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    eng.step(args.steps)
    eng.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    power = sampler.stop() if sampler else None
    barrier()
    conv_ms, conv_flop, conv_n = eng.profile_conv_read()
    eng.profile_conv(False)
    search_ms, search_steps = eng.profile_search_read()
    eng.profile_search(False)
    s1 = eng.stats()

    # N > 1: rank 0 repeats the K steps ALONE (every other rank waits at the barrier behind it), so that the line carries
    # the N = 1 rate of the same box and `weak_scaling_efficiency` needs no second run
    solo = None
    if world > 1:
        if rank == 0:
            eng.sync()
            q0 = eng.stats()
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            eng.step(args.steps)
            eng.sync()
            torch.cuda.synchronize()
            ts1 = time.perf_counter()
            solo = {"positions_per_s": (eng.stats()["positions"] - q0["positions"]) / (ts1 - ts0), "seconds": ts1 - ts0,
                    "what": f"rank 0 alone on its GPU, the same {args.steps} steps, the other ranks idle at a barrier"}
        barrier()

    # Extra leg, reported beside `value`, never as it: the same K steps with the Winograd operands carried as two f16
    # halves (AGZ_PRECISION_F32S: f32 network, f32 accumulate, fp16 MFMA; agrees with the float64 oracle as closely
    # as the exact-f32 path does, tests/test_gpu_nn32s.py).  N = 1 only, exact-f32 runs only.
    alt = None
    if world == 1 and args.precision == "f32" and not args.no_alt_precision:
        try:
            eng.set_precision("f32s")
            eng.step(args.warmup)
            eng.sync()
            a0 = eng.stats()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            eng.step(args.steps)
            eng.sync()
            torch.cuda.synchronize()
            tb_ = time.perf_counter()
            a1 = eng.stats()
            alt = {"precision": "f32s: f32 network, Winograd GEMM operands as two f16 halves on v_mfma_f32_32x32x16_f16, f32 accumulate",
                   "value": (a1["positions"] - a0["positions"]) / (tb_ - ta), "unit": "positions/s",
                   "ms_per_step": 1e3 * (tb_ - ta) / args.steps, "steps": args.steps,
                   "parity": "max |d pi|, |d v| vs the float64 oracle <= 1e-6 on the configs of tests/test_gpu_nn32s.py (bar 1e-4)"}
            eng.set_precision("f32")
        except Exception as ex:
            alt = {"error": str(ex)}

    elapsed = t1 - t0
    d = {k: s1[k] - s0[k] for k in ("positions", "evals", "duplicate_evals", "terminal_visits", "root_visits",
                                    "games_finished", "steps")}
    per_rank = None
    if dist is not None:
        mine = torch.tensor([d["positions"], elapsed, conv_ms / max(conv_n, 1)], dtype=torch.float64, device=rdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "positions": float(t[0]), "seconds": float(t[1]), "positions_per_s": float(t[0] / t[1]),
                     "layer_ms": float(t[2])} for r, t in enumerate(x.tolist() for x in allr)]
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([d["positions"], d["evals"], d["root_visits"], d["games_finished"]], dtype=torch.float64,
                         device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        d["positions"], d["evals"], d["root_visits"], d["games_finished"] = [float(x) for x in c.tolist()]
    # The one exchange step of the path, outside the timed region: every rank's finished records
    # are all-gathered into every rank's device replay arena by libagz itself (agz_allgather_records: RCCL over
    # xGMI, device to device).  torch.distributed only carries the 128-byte RCCL unique id to the other ranks.
    # SURVEY.md 8d's own definition, on request: games played to their natural end (selfplay.jl:22-43), slots recycled
    generation = None
    if args.generation > 0 and world == 1:
        G = args.generation
        import numpy as np
        cap_warm, cap_win = args.generation_seconds / 2.0, args.generation_seconds / 2.0
        g0 = eng.stats()["games_finished"]
        tw = time.perf_counter()
        while eng.stats()["games_finished"] - g0 < G and time.perf_counter() - tw < cap_warm:
            eng.step(25)
        warm_s = time.perf_counter() - tw
        warm_games = eng.stats()["games_finished"] - g0
        eng.records_clear()
        q0 = eng.stats()
        gsampler = PowerSampler(local_rank).start()
        tg0 = time.perf_counter()
        while True:
            eng.step(25)
            q1 = eng.stats()                       # synchronises
            if q1["games_finished"] - q0["games_finished"] >= G or time.perf_counter() - tg0 > cap_win:
                break
        wall = time.perf_counter() - tg0
        gpower = gsampler.stop()
        recs = eng.records()
        nm = np.array([r["num_moves"] for r in recs], np.int64)
        gd = {k: q1[k] - q0[k] for k in ("positions", "evals", "games_finished", "resigned_games", "steps", "terminal_visits")}
        generation = {
            "what": f"window in which {G} games ended naturally (resignation, two passes or the move limit), after an untimed "
                    f"warm-up in which {G} others did; slots recycled throughout",
            "generation_rate": float(nm.sum()) / wall, "steady_state_rate": gd["positions"] / wall, "unit": "positions/s",
            "generation_positions": int(nm.sum()), "steady_state_positions": gd["positions"], "wall_s": wall,
            "steps": gd["steps"], "ms_per_step": 1e3 * wall / max(gd["steps"], 1), "games_finished": gd["games_finished"],
            "resigned_games": gd["resigned_games"], "records_read": len(recs), "records_dropped": q1["records_dropped"],
            "game_length": {"mean": float(nm.mean()), "min": int(nm.min()), "max": int(nm.max())} if len(nm) else None,
            "evals_per_position": gd["evals"] / max(gd["positions"], 1), "terminal_visits": gd["terminal_visits"],
            "batch_fill": gd["evals"] / max(gd["steps"] * 8 * args.games, 1), "warmup_s": warm_s,
            "warmup_games_finished": warm_games, "requested_games": G,
            "capped": bool(warm_games < G or gd["games_finished"] < G), "cap_seconds": args.generation_seconds,
            "power": gpower,
            "note": "generation_rate counts the moves of the games that ENDED in the window (8d), steady_state_rate the moves "
                    "PLAYED in it (bench.py's `value` definition).  With the default G = games per GPU the warm-up is a whole "
                    "generation (every game of the staggered start-up population has been replaced) and the window is the next "
                    "one: SURVEY.md 8d's 'timing over >= 1 full generation after a warm-up generation'.  A capped or smaller "
                    "window holds the shortest games first and under-reads",
        }
        eng.records_clear()

    exchange = None
    exchange_hung = False
    if dist is not None:
        # the timed region of the headline config sees no game end (2 of ~76 moves per game); the exchange needs
        # finished games: keep playing, untimed, until this rank has some (bounded)
        # (>= 16 games per rank, so that the payload collective moves unequal, non-trivial chunks: VERDICT r3 #6)
        for _ in range(60):
            if eng.records_count() >= 16:
                break
            eng.step(50)
        # RCCL prints a version banner on stdout when a communicator is created; stdout must carry exactly one
        # JSON line, so the exchange leg runs with fd 1 pointed at stderr.  It also runs on a watchdog: a
        # collective that never completes (a rank missing, a fabric problem) must not swallow the self-play number.
        # --single-device-test (every rank on cuda:0, where RCCL refuses two ranks on one device) runs the SAME leg
        # with gloo carrying the two collectives: pack, count check / stride (agz_gather_plan), payload, device-side
        # indexing and compaction into the arena are the C ABI's either way (Engine.allgather_records_hosted).
        import threading
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        box = {}

        def _exchange():
            try:
                own = eng.records_count()
                own_bytes = eng.records_packed_size()
                if args.single_device_test:
                    barrier()
                    e0 = time.perf_counter()
                    added = eng.allgather_records_hosted()
                    how = ("host-carried: agz_records_export_packed -> gloo all_gather (counts, then payload padded by "
                           "agz_gather_plan) -> agz_replay_ingest_gathered")
                else:
                    ids = [ag.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    comm = eng.comm_create(rank, world, ids[0])
                    barrier()
                    e0 = time.perf_counter()
                    added = eng.allgather_records(comm)
                    how = "agz_allgather_records (count exchange + padded in-place ncclAllGather, device to device)"
                eng.sync()
                e1 = time.perf_counter()
                # every rank must now hold the same arena: sum of everybody's finished games, rank order
                chk = torch.tensor([own, added, eng.replay_positions(), own_bytes], dtype=torch.int64, device=rdev)
                allc = [torch.zeros_like(chk) for _ in range(world)]
                dist.all_gather(allc, chk)
                allc = [[int(v) for v in t.tolist()] for t in allc]
                box["ok"] = {"collective": how, "games_in_arena": added, "positions_in_arena": eng.replay_positions(),
                             "ms": 1e3 * (e1 - e0), "bytes": sum(c[3] for c in allc), "bytes_by_rank": [c[3] for c in allc],
                             "gb_s_into_each_arena": sum(c[3] for c in allc) / max(e1 - e0, 1e-9) / 1e9,
                             "own_games": own, "own_games_by_rank": [c[0] for c in allc],
                             "consistent": all(c[1] == sum(x[0] for x in allc) and c[2] == allc[0][2] for c in allc)}
                eng.records_clear()
                if not args.single_device_test:
                    eng.comm_destroy(comm)
            except Exception as ex:      # the exchange leg must never take the self-play number down with it
                box["err"] = f"{type(ex).__name__}: {ex}"

        th = threading.Thread(target=_exchange, daemon=True)
        th.start()
        th.join(timeout=120.0)
        exchange_hung = th.is_alive()
        exchange = box.get("ok") or {"error": box.get("err", "timed out after 120 s")}
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)

    if s1["stalled_games"] or s1["pool_short_searches"]:
        # the default pool holds every tree of these workloads with room to spare (`pool` in the line says how much); a
        # shortened or waiting search here means the number is not the reference's workload
        raise SystemExit(f"node pool: {s1['pool_short_searches']} shortened searches, {s1['stalled_games']} waiting games "
                         f"(capacity {s1['node_capacity']}): results invalid")

    if rank == 0:
        value = d["positions"] / elapsed
        fpos = R * f_eval(N, tower)
        T = (N + 2) // 3
        f16, f32s = args.precision == "f16", args.precision == "f32s"
        # executed / algorithmic multiplies: F(3x3,3x3) needs 25 per 3x3 tile; boards of 13x13 and larger run the exact-f32
        # tower on F(4x4,3x3) (agz_wino4.hip): 36 per 4x4 tile
        f43 = (not f16) and (not f32s) and N >= 13 and args.winograd == 1
        T4 = (N + 3) // 4
        wino_ratio = 1.0 if f16 else (36.0 * T4 * T4 / (9.0 * N * N) if f43 else 25.0 * T * T / (9.0 * N * N))
        peak = 2500.0 if f16 else PEAK_F32_MFMA_TFLOPS
        rows_per_launch = conv_flop / max(conv_n, 1) / (2.0 * 9 * 256 * 256)
        traffic_committed = None
        traffic, traffic_src = pmc_traffic(rows_per_launch, N, args.precision, f43)
        if world == 1 and headline and args.stagger > 0 and not args.no_live_traffic:
            # driver-witnessed: counters collected on THIS box, now; the committed file is the fallback
            try:
                bpr, src = live_pmc_traffic(args)
            except Exception as ex:
                bpr, src = None, f"{type(ex).__name__}: {ex}"
            if bpr is not None:
                traffic_committed = traffic
                traffic, traffic_src = bpr * rows_per_launch, src
            else:
                traffic_src = f"{traffic_src}; live passes unavailable ({src})"
        # The roofline object.  `achieved`/`frac` are what the MFMA pipe EXECUTES: Winograd F(3x3,3x3) needs 25
        # multiplies per 3x3 output tile and (cin, cout) pair instead of 81, all of them still f32, so the
        # honest fraction of the f32 MFMA peak is executed flops / time / peak (<= 1).  The rate in terms of the
        # direct convolution's flops (SURVEY.md 8d's per-unit figure) is reported next to it as
        # `achieved_algorithmic`; it may exceed the peak by up to the strength-reduction factor.
        conv_s = conv_ms * 1e-3
        alg_tf = conv_flop / conv_s / 1e12 if conv_ms > 0 else None
        exe_tf = alg_tf * wino_ratio if alg_tf is not None else None
        if f32s:
            # split-operand form: the matrix pipe is 4x faster than the f32 MFMA it replaces and stops being the bound;
            # the layer is bound by moving V (the 25 transformed planes) through HBM: measured traffic / launch time
            # against the 8 TB/s HBM peak.  `achieved_algorithmic` stays the direct convolution's flops / time.
            tb = traffic / (conv_ms * 1e-3 / max(conv_n, 1)) / 1e9 if traffic else None
            roofline = {
                "bound": "hbm", "kernel": "3x3 256->256 tower conv (Winograd F(3x3,3x3), operands as two f16 halves on "
                                          "v_mfma_f32_32x32x16_f16, f32 accumulate)",
                "achieved": tb, "peak": 8000.0, "unit": "GB/s", "frac": tb / 8000.0 if tb else None,
                "traffic": traffic, "traffic_source": traffic_src, "achieved_algorithmic": alg_tf,
                "launches_timed": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
                "note": "traffic = HBM bytes per layer from the PMC passes named in traffic_source",
            }
        roofline = roofline if f32s else {
            "bound": "mfma",
            "kernel": "3x3 256->256 tower conv = k_conv3x3_f16 (implicit GEMM, v_mfma_f32_32x32x16_f16)" if f16 else
                      ("3x3 256->256 tower conv (Winograd F(4x4,3x3) in six one-row passes on v_mfma_f32_32x32x2_f32; every kernel of a layer timed together)" if f43 else
                       "3x3 256->256 tower conv (Winograd F(3x3,3x3) on v_mfma_f32_32x32x2_f32; every kernel of a layer timed together)"),
            "achieved": exe_tf, "peak": peak, "unit": "TFLOP/s",
            "frac": exe_tf / peak if exe_tf is not None else None,
            "achieved_algorithmic": alg_tf,
            "algorithmic_over_executed": 1.0 / wino_ratio,
            "note": "achieved = flops the MFMA pipe executes per launch / average launch time (HIP events on the engine's "
                    "stream around every tower-conv launch of the timed region); achieved_algorithmic = 2*rows*9*256*256 "
                    "per launch / the same time",
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_from_committed_profile": traffic_committed,      # (set when `traffic` was measured live: the file's figure beside it)
            "algorithmic_bytes_per_launch": (1280.0 if f16 else 2560.0) * conv_flop / max(conv_n, 1) / (2.0 * 9 * 256 * 256),
            "launches_timed": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
            "executed_flop_per_launch_avg": conv_flop * wino_ratio / max(conv_n, 1),
            "algorithmic_flop_per_launch_avg": conv_flop / max(conv_n, 1),
        }
        if not f16 and not f32s and exe_tf is not None and not args.no_sustained:
            # context, not the contract's `peak`: the constant-operand MFMA-only figure of rounds 3-4 (bounded by its own loop), measured
            # now, after the timed region (0.4 s of back-to-back register-only MFMA launches; HISTORY.md 4f)
            try:
                sus = eng.mfma_sustained_tflops(400)
                roofline["sustained_mfma"] = {
                    "value": sus, "unit": "TFLOP/s", "frac_of_nominal_peak": sus / peak, "achieved_over_sustained": exe_tf / sus,
                    "what": "v_mfma_f32_32x32x2_f32 from registers only (two CONSTANT operands, a loop of 8 MFMAs), one launch of "
                            "~10 ms after another for 0.4 s on this GPU, median of the second half.  Kept for continuity with "
                            "rounds 3-4; it is bounded by its own loop (610-700 W at 2.3 GHz), not by the board: see "
                            "sustained_mfma_changing_operands",
                }
                # the same with operands that change in front of every MFMA (dense pseudo-random A and B, as Winograd-transformed
                # operands are) and a loop unrolled 16 x 8 deep: THIS is the ceiling a real layer has.  (Round 5, HISTORY.md 12:
                # the constant-operand figure above is bounded by its own 8-MFMA loop, not by the board's power.)
                susd = eng.mfma_sustained_data_tflops(400, 3)
                roofline["sustained_mfma_changing_operands"] = {
                    "value": susd, "unit": "TFLOP/s", "frac_of_nominal_peak": susd / peak, "achieved_over_sustained": exe_tf / susd,
                    "what": "v_mfma_f32_32x32x2_f32 from registers only, every MFMA a different pair of 16 dense pseudo-random A / B "
                            "register values per lane, 128 MFMAs per loop iteration",
                }
            except Exception as ex:      # never at the expense of the line
                roofline.setdefault("sustained_mfma", {"error": f"{type(ex).__name__}: {ex}"})
        out = {
            "metric": f"self-play positions/sec ({N}x{N}, tower={tower}, {R} readouts)" + (" [fp16 tower]" if f16 else "")
                      + (" [f32 as split f16 operands]" if f32s else ""),
            "value": value, "unit": "positions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if f16 else "f32 (Winograd operands as 2 x f16, f32 accumulate)" if f32s else "f32", "data": "synthetic",
            "config": {
                "workload": f"GoEnv({N}), tower_height={tower}, {R} readouts, {args.games} concurrent games per GPU, "
                            f"8 leaves per game per step (batch <= {8 * args.games} positions)",
                "games_per_gpu": args.games, "parallel_readouts": 8, "stagger_moves": args.stagger,
                "setup_prelude_steps": prelude,
                "weights": "synthetic glorot-uniform (seed 0), BN identity", "parallelism": f"games sharded x{world}",
            },
            "positions": d["positions"], "evals": d["evals"],
            "evals_per_position": d["evals"] / max(d["positions"], 1),
            "duplicate_evals": d["duplicate_evals"], "terminal_visits": d["terminal_visits"],
            "readout_positions_per_s": d["root_visits"] / R / elapsed,
            "games_finished": d["games_finished"],
            "end_to_end_algorithmic_tflops": value * fpos / world / 1e12,          # positions/s x F_position
            "end_to_end_executed_mfma_frac": value * fpos * wino_ratio / (world * peak * 1e12),
            "roofline": roofline,
            "power": power,        # socket power and shader clock over the timed region (PowerSampler)
            "pool": {"node_capacity": s1["node_capacity"], "peak_nodes_per_game": s1["peak_nodes_per_game"],
                     "short_searches": s1["pool_short_searches"], "refused_allocations": s1["pool_exhausted"]},
        }
        if power and power.get("sclk_mhz") and not f16 and not f32s and exe_tf is not None:
            # what the nominal peak becomes at the clock the timed region actually ran at (the peak assumes 2.4 GHz)
            mhz = power["sclk_mhz"]["mean"]
            roofline["at_measured_clock"] = {"sclk_mhz": mhz, "peak": peak * mhz / 2400.0,
                                             "frac": exe_tf / (peak * mhz / 2400.0), "unit": "TFLOP/s",
                                             "note": "peak scaled from the 2.4 GHz nominal clock to the mean sampled sclk"}
        if search_steps:
            out["search_kernels"] = search_kernels_object(search_ms, search_steps, d if world == 1 else {**d, "evals": d["evals"] / world},
                                                          args.games, N, 1e3 * elapsed / args.steps, live_pmc_traffic.search)
        if generation is not None:
            out["generation"] = generation
            if not generation["capped"]:
                # SURVEY.md 8d's definition of the metric is the generation rate (sum of position.n of the games that ENDED in a
                # whole generation after a warm-up generation / wall time): when that leg ran to its end it IS `value`
                # (VERDICT r5 #4); the K-step window stays in the line as `steady_state` and behind ms_per_step / steps / warmup
                out["steady_state"] = {"value": value, "unit": "positions/s", "steps": args.steps, "ms_per_step": 1e3 * elapsed / args.steps,
                                       "what": "moves that completed their full readout budget inside the K timed steps / their time "
                                               "(every game mid-search at a random phase; no game ends in the window)"}
                out["value"] = generation["generation_rate"]
                out["value_definition"] = ("generation.generation_rate: sum of the lengths of the games that ended in the measured "
                                           "generation / its wall time (SURVEY.md 8d; selfplay.jl:22-43), on the same engine right behind "
                                           "the K timed steps; ms_per_step, steps, warmup and roofline describe the K-step window "
                                           "(`steady_state`)")
                out["end_to_end_algorithmic_tflops"] = out["value"] * fpos / world / 1e12
                out["end_to_end_executed_mfma_frac"] = out["value"] * fpos * wino_ratio / (world * peak * 1e12)
        if per_rank is not None:
            out["per_rank"] = per_rank      # each rank's own clock over the same K steps (`value` uses the slowest)
            if solo is not None:
                out["single_rank_same_box"] = solo
                out["weak_scaling_efficiency"] = min(r["positions_per_s"] for r in per_rank) / solo["positions_per_s"]
                if args.single_device_test:
                    out["weak_scaling_note"] = (f"--single-device-test: all {world} ranks share ONE GPU, so ~1/{world} is what this "
                                                "field should read here; it means something on one GPU per rank")
        if exchange is not None:
            out["exchange"] = exchange
        if alt is not None:
            out["alt_precision"] = alt
        if world == 1 and headline and args.stagger > 0 and not args.no_config_legs:
            # BASELINE configs[3] / configs[4], one GPU's shard each, driver-witnessed: never `value`
            for key, cfg in (("config3", ("configs[3]", 19, 20, 800, 256, "f32")), ("config4", ("configs[4]", 19, 20, 1600, 512, "f16"))):
                try:
                    out[key] = shard_leg(ag, torch, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], args.config_leg_steps,
                                         5, 60, local_rank)
                except Exception as ex:      # never at the expense of the line
                    out[key] = {"config": cfg[0], "error": f"{type(ex).__name__}: {ex}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, tower, R, args.cpu_baseline_seconds)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "positions/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), file=line_out, flush=True)
    if exchange_hung:          # a stuck collective also blocks engine teardown: the line is out, leave
        sys.stdout.flush()
        os._exit(0)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
