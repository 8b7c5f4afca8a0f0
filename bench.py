#!/usr/bin/env python3
"""Throughput bench for the MI355X decode-DSP hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload aac_synth|fir|pipeline]

One rank per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks
of the timed region: the streams are independent, so there is no data-path collective
and scaling is weak -- every rank decodes its own 4096-stream batch).

A step = one pass of the hot path over one batch of synthetic, device-resident input:
  pipeline  : (default) 4096 streams x 64 frames of 48 kHz stereo spectra -> IMDCT + window + OLA ->
              48k->16k MFMA FIR -> interleaved s16: the worker's whole device-side tail, the
              "decode + resample on a 4096-stream batch" the metric is quoted on
  aac_synth : the IMDCT + window + OLA kernel alone (BASELINE configs[1] at the metric's batch size)
  fir       : BASELINE configs[2]: 4096 streams x 2 ch x 1 s of 48 kHz f32 -> 16 kHz
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r04_pmc.json")  # rocprofv3 --pmc passes of this same workload (tools/profile_bench.sh)
PMC_KERNEL = {"aac_synth_s16out": "k_aac_synth", "aac_synth": "k_aac_synth_f32out", "fir_pipeline_s16in": "k_fir_48k_16k"}


def pmc_traffic(kind, **config):
    """HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KB: tools/summarize_pmc.py), if
    they were taken on exactly this workload; otherwise None."""
    try:
        summary = json.load(open(PMC_SUMMARY))
        entry = summary[PMC_KERNEL[kind]]
        took = summary["_bench_line_under_profiler"]["config"]
    except (OSError, KeyError, ValueError):
        return None
    rows = took["streams_per_gpu"] * took["channels"]
    same = (config == {"streams": took["streams_per_gpu"], "frames": took["frames_per_stream"], "channels": took["channels"]} or
            config == {"rows": rows, "frames": took["frames_per_stream"] * 1024})
    return entry["traffic_bytes"] if same else None

MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 matrix peak
SEED0 = 0x12345678
# the reference's test spectrum (+-12) synthesises to well under one 16-bit step; at this gain the 16 kHz s16 output peaks at
# a few hundred steps (tests/test_pipeline_gpu.py checks the same data against the oracle).  Not a performance knob:
# nothing on the path is data-dependent.
SPECTRUM_GAIN = 2500.0


def lcg_tables(n=1024):
    """Jump-ahead constants of the reference's test LCG (dsp.rs:725-738): state_k = A_k*seed + C_k mod 2^32."""
    a, c, m = 1664525, 1013904223, 1 << 32
    A, Cc = np.zeros(n, np.int64), np.zeros(n, np.int64)
    ak, ck = 1, 0
    for k in range(n):
        ak, ck = (ak * a) % m, (ck * a + c) % m
        A[k], Cc[k] = ak, ck
    return A, Cc


def seeded_spectra(torch, device, streams, frames, ch, stream0=0):
    """[streams*frames][ch][1024] f32 on device: the reference's seeded_spectrum (dsp.rs:725-738) with
    seed = 0x12345678 + stream*0x9e3779b9 + frame*2 + channel; +-12, every 7th bin zero."""
    A, Cc = lcg_tables()
    A = torch.from_numpy(A).to(device)
    Cc = torch.from_numpy(Cc).to(device)
    s = torch.arange(stream0, stream0 + streams, device=device, dtype=torch.int64).view(-1, 1, 1)
    f = torch.arange(frames, device=device, dtype=torch.int64).view(1, -1, 1)
    c = torch.arange(ch, device=device, dtype=torch.int64).view(1, 1, -1)
    seed = (SEED0 + s * 0x9E3779B9 + f * 2 + c) & 0xFFFFFFFF
    out = torch.empty((streams * frames, ch, 1024), dtype=torch.float32, device=device)
    keep = (torch.arange(1024, device=device) % 7 != 0).to(torch.float32)
    rows = seed.reshape(-1, ch)
    step = max(1, (1 << 22) // (ch * 1024))
    for i in range(0, rows.shape[0], step):
        st = (rows[i:i + step].unsqueeze(-1) * A + Cc) & 0xFFFFFFFF
        v = ((st >> 8) & 0xFFFF).to(torch.float32) / 32768.0 - 1.0
        out[i:i + step] = v * 12.0 * keep
    return out


def cpu_baseline_synth(target_s=12.0):
    """The oracle (CPU restatement, 'port'), single thread, on a bounded sample of the same workload."""
    from oracle import oracle as O
    n_streams, n_frames = 2, 64
    spectra = np.stack([[[O.seeded_spectrum(1024, (SEED0 + s * 0x9E3779B9 + f * 2 + c) & 0xFFFFFFFF) for c in range(2)]
                         for f in range(n_frames)] for s in range(n_streams)])
    chans = [[O.Channel(), O.Channel()] for _ in range(n_streams)]
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        for s in range(n_streams):
            for f in range(n_frames):
                for c in range(2):
                    chans[s][c].synthesize(spectra[s, f, c], 0, f & 1)
                done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "oracle sko_synthesize_channel, %d stereo frames (2 streams x 64 frames looped) in %.1f s" % (done, dt)}


def cpu_baseline_pipeline(target_s=15.0):
    """Oracle ('port'), single thread: the same chain on one stream x 64 frames, looped for ~15 s."""
    from oracle import oracle as O
    n_frames = 64
    spectra = np.stack([[O.seeded_spectrum(1024, (SEED0 + f * 2 + c) & 0xFFFFFFFF) for c in range(2)] for f in range(n_frames)])
    spectra = (spectra * np.float32(SPECTRUM_GAIN)).astype(np.float32)
    chans = [O.Channel(), O.Channel()]
    shapes = [[f & 1, f & 1] for f in range(n_frames)]
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        pcm, chans = O.synthesize_stream(spectra, [[0, 0]] * n_frames, shapes, chans)
        planar = np.ascontiguousarray(pcm.transpose(1, 0, 2).reshape(2, n_frames * 1024))
        y = O.downsample_planar(planar, 48000, 16000)
        O.planar_f32_to_s16_interleaved(y)
        done += n_frames
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "oracle synth + sko_downsample_planar + s16 interleave, %d stereo frames (1 stream x 64 frames looped) in %.1f s" % (done, dt)}


def usable_cores():
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline_all_cores(workload, target_s=8.0):
    """The oracle loop in one process per usable core (affinity and cgroup quota), started as child processes before
    this process touches the GPU; the sum of their rates.  The children load a build of oracle/sk_oracle.c made here,
    for this host's cores (-O3 -march=native, -ffp-contract=off kept: same results, tests/test_oracle_pins.py), when a
    compiler is present; else the portable -O2 library."""
    import subprocess
    import tempfile
    from oracle import oracle as O
    cores = usable_cores()
    tmp = tempfile.mkdtemp(prefix="sk_oracle_native_")
    native = O.build_native(tmp)
    env = dict(os.environ)
    if native:
        env["SK_ORACLE_LIB"] = native
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench.CPU_BASELINES[%r](%f)))" % (ROOT, workload, target_s))
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
             for _ in range(cores)]
    results = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode == 0 and out.strip():
            results.append(json.loads(out.strip().splitlines()[-1]))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if not results:
        return None
    return {"value": sum(r["value"] for r in results), "unit": results[0]["unit"], "cores": len(results), "kind": "port",
            "build": "-O3 -march=native -ffp-contract=off" if native else "-O2 -ffp-contract=off (no compiler on this host)",
            "sample": "%d processes, each: %s" % (len(results), results[0]["sample"])}


def cpu_baseline_fir(target_s=10.0):
    from oracle import oracle as O
    x = np.random.default_rng(0).uniform(-1, 1, (2, 48000)).astype(np.float32)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        O.downsample_planar(x, 48000, 16000)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "stream-seconds/s", "cores": 1, "kind": "port",
            "sample": "oracle sko_downsample_planar, %d x 1 s of 48 kHz stereo in %.1f s" % (done, dt)}


def cpu_baseline_pcm(target_s=10.0):
    """Oracle ('port'), single thread: float_sample_to_i16 (soundkit-decoder lib.rs:1815-1827) over 16 Mi samples, looped."""
    from oracle import oracle as O
    rng = np.random.default_rng(SEED0)
    x = (rng.random(1 << 24, dtype=np.float32) * 2.2 - 1.1).astype(np.float32)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        O.pcm_convert("FLOAT_TO_I16_ROUND", x)
        done += x.size
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "sko_pcm_convert FLOAT_TO_I16_ROUND, %d samples (16 Mi looped) in %.1f s" % (done, dt)}


E2E_CLIP = "aac-stereo-48k.adts"


def cpu_chain_decode(clip_bytes, loops, out_rate=16000, mono=True):
    """The whole decode of one ADTS stream on the CPU, in the worker's order (soundkit-decoder lib.rs:1793-1813, 3324-3456; the
    loop to mirror: aac-wasm-bench/src/lib.rs:1589-1611): ADTS framing + the product's HOST front-end (csrc/aac_frontend.cpp:
    Huffman, stereo tools, TNS, PNS -- plain C++, no GPU) -> oracle synthesis -> float_sample_to_i16 -> / 32768 -> oracle
    streaming resampler -> mono downmix -> s16.  -> (s16 mono samples, access units decoded).  cpu_baseline leg only."""
    from oracle import oracle as O
    from soundkit_amd import aac_lc
    frames = aac_lc.split_adts(clip_bytes)
    fe = aac_lc.AacLcFrontEnd(frames[0][0])
    chans = [O.Channel() for _ in range(fe.channels)]
    rs = O.StreamingResampler(fe.sample_rate, out_rate, fe.channels) if fe.sample_rate != out_rate else None
    out, units = [], 0

    def tail(planar):
        if planar.shape[1]:
            out.append(O.planar_f32_to_s16_interleaved(O.downmix_mono(planar)[None] if mono and planar.shape[0] > 1 else planar))
    for _ in range(loops):
        parsed = [fe.parse(au) for _, au in frames]
        coeffs = np.stack([c for c, _, _ in parsed])
        pcm, _ = O.synthesize_stream(coeffs, [sq for _, sq, _ in parsed], [sh for _, _, sh in parsed], chans)
        for f in range(pcm.shape[0]):  # one AudioData per access unit enters the resampler, as in the worker
            q = O.planar_f32_to_s16_interleaved(pcm[f]).reshape(1024, fe.channels).T.astype(np.float32) / np.float32(32768.0)
            tail(rs.process(q) if rs else q)
        units += len(frames)
    if rs:
        tail(rs.flush())
    fe.close()
    return (np.concatenate([o.reshape(-1) for o in out]) if out else np.zeros(0, np.int16)), units


def cpu_baseline_decode(target_s=12.0):
    """Whole-decode CPU baseline ('port'), single thread: cpu_chain_decode on the end_to_end clip, looped for ~target_s."""
    clip = open(os.path.join(ROOT, "tests", "golden", "aac", E2E_CLIP), "rb").read()
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_s:
        _, units = cpu_chain_decode(clip, 4)
        done += units
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "host front-end (product C++: ADTS framing, Huffman, stereo tools, TNS) + oracle synthesis + s16 + oracle streaming "
                      "48 -> 16 kHz resampler + mono + s16 per access unit, %d access units (%s looped) in %.1f s" % (done, E2E_CLIP, dt)}


def pcm_stats_s16(lg, samples):
    """PcmStats::from_pcm (aac-wasm-bench/src/lib.rs:66-101) of s16 samples as f32 = s / 32768, by the load generator's C helper"""
    import ctypes as C
    samples = np.ascontiguousarray(samples, np.int16)
    rms, peak, fnv = C.c_double(0), C.c_double(0), C.c_uint64(0)
    lg.sk_loadgen_pcm_stats.restype = None
    lg.sk_loadgen_pcm_stats.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lg.sk_loadgen_pcm_stats(samples.ctypes.data, samples.size, C.byref(rms), C.byref(peak), C.byref(fnv))
    return {"sample_count": int(samples.size), "rms": rms.value, "peak_abs": peak.value, "fnv1a": "0x%016x" % fnv.value}


CPU_BASELINES = {"decode": cpu_baseline_decode, "pcm": cpu_baseline_pcm, "fir": cpu_baseline_fir, "aac_synth": cpu_baseline_synth, "pipeline": cpu_baseline_pipeline}


class EndToEndStalled(RuntimeError):
    """the load generator's progress deadline fired; its record is on stderr"""


def end_to_end(args, eng, torch, dist, world, rank, device, emit=True, cpu_baseline=None):
    """SURVEY 8d config 5, scaled: `--streams` ADTS AAC-LC streams (the 48 access units of the reference's 48 kHz
    stereo TS sample, looped) through the batch scheduler: host entropy decode -> GPU ticks -> 16 kHz mono s16 out.
    One step = one pass of the clip through every stream.  Everything is inside the timed region: framing, Huffman,
    H2D, synthesis, resampling, packing, D2H and delivery."""
    import ctypes as C
    from soundkit_amd import pipeline, sharding
    from soundkit_amd._lib import DecodeOptionsC
    from soundkit_amd import aac_lc
    is_mp3 = args.clip.endswith(".mp3")
    unit_frames = 576 if is_mp3 else 1024  # PCM frames per unit: an MP3 granule / an AAC access unit
    if is_mp3:  # an MPEG Layer III stream: the scheduler picks the decoder from the first bytes; a unit = one granule
        from soundkit_amd import mp3 as mp3_mod
        clip = open(os.path.join(ROOT, "tests", "golden", "mp3", args.clip), "rb").read()
        found, used = mp3_mod.scan(clip)
        clip = clip[found[0].offset:used]  # without the ID3 tag: the clip is looped
        units = sum(f.granules for f in found)
        src_rate, src_ch = found[0].sample_rate, found[0].channels
    else:
        clip = open(os.path.join(ROOT, "tests", "golden", "aac", args.clip), "rb").read()
        adts = aac_lc.split_adts(clip)
        units = len(adts)
        clip = clip[:sum(len(au) + 7 for _, au in adts)]  # whole frames only: the clip is looped
        src = aac_lc.AacLcFrontEnd(adts[0][0])
        src_rate, src_ch = src.sample_rate, src.channels
        src.close()
    lg = C.CDLL(os.path.join(ROOT, "soundkit_amd", "libsk_loadgen.so"))

    class Result(C.Structure):
        _fields_ = [("seconds", C.c_double)] + [(n, C.c_uint64) for n in ("access_units", "outputs", "pcm_frames", "pcm_bytes", "errors", "input_full")]
    lg.sk_loadgen_run.restype = C.c_int
    lg.sk_loadgen_run.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                  C.c_uint32, C.c_void_p]

    class Check(C.Structure):  # sk_load_check; hash = NULL: capture only (no per-stream hashing in the timed region)
        _fields_ = [("hash", C.c_void_p), ("outputs", C.c_void_p), ("bytes", C.c_void_p), ("errors", C.c_void_p), ("capture", C.c_void_p),
                    ("n_capture", C.c_uint32), ("capture_buf", C.c_void_p), ("capture_cap", C.c_size_t), ("capture_len", C.c_void_p)]
    lg.sk_loadgen_run_checked.restype = C.c_int
    lg.sk_loadgen_run_checked.argtypes = lg.sk_loadgen_run.argtypes + [C.c_void_p]
    # stream 0 of the first generator is kept whole: the line carries the PcmStats of what was decoded
    # (aac-wasm-bench/src/lib.rs:66-101, 513-550: the reference's harness prints them beside every rate)
    out_ch = args.out_channels or src_ch
    cap_bytes = int(units * (args.steps + args.warmup + 1) * unit_frames * 2 * out_ch * ((args.out_rate or src_rate) / src_rate + 0.01)) + (1 << 16)
    cap_index = np.zeros(1, np.uint32)
    cap_buf, cap_len = np.zeros(cap_bytes, np.uint8), np.zeros(1, np.uint64)
    chk = Check(None, None, None, None, cap_index.ctypes.data, 1, cap_buf.ctypes.data, cap_bytes, cap_len.ctypes.data)
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    share = max(4, min(cores, 16))  # the pool gives a one-GPU job 16 cores; with several ranks `cores` is already this rank's share
    threads = args.entropy_threads or max(1, share - 3 - args.feeders)
    feeders = args.feeders
    import threading
    import soundkit_amd
    front_mode = {"host": 0, "gpu": 1, "quant": 2}[args.front_end] if args.front_end else int(args.gpu_entropy)
    n_sched = max(1, args.schedulers)
    n_gen = n_sched  # one load generator (its own producer and consumer threads) per scheduler: wait_outputs is per pipeline
    per_streams = args.streams // n_sched
    engines = [eng] + [soundkit_amd.Engine(eng.device if hasattr(eng, "device") else 0, max(per_streams, 16)) for _ in range(n_sched - 1)]
    scheds = [pipeline.BatchScheduler(engines[i], entropy_threads=max(1, threads // n_sched), max_streams=per_streams,
                                      max_frames_per_tick=args.tick_frames, max_stream_frames_per_tick=args.stream_frames_per_tick,
                                      gpu_entropy=front_mode, tick_wait_us=args.tick_wait_us, lanes=args.lanes) for i in range(n_sched)]
    opt = DecodeOptionsC(args.out_rate, 16, args.out_channels, 0)

    class Summed:
        pass

    def run(loops):
        results = [Result() for _ in range(n_gen)]
        rcs = [0] * n_gen

        def one(i):
            rcs[i] = lg.sk_loadgen_run_checked(scheds[i % n_sched]._h, clip, len(clip), units, args.streams // n_gen, loops, C.byref(opt),
                                               max(2, feeders // n_gen), 0, C.byref(results[i]), C.byref(chk) if i == 0 else None)
        ths = [threading.Thread(target=one, args=(i,)) for i in range(n_gen)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if any(rc == -8 for rc in rcs):
            # the load generator's progress deadline: its record (the scheduler's threads, batches, stream table and the
            # engine's stage) is on stderr.  Leave at once, without tearing the pipeline down: a device that stopped
            # answering would hold the teardown too, and the record is what matters.
            sys.stderr.write("bench.py: the end-to-end run stalled (SK_ERR_TIMEOUT from the load generator); state dumped above\n")
            sys.stderr.flush()
            if not emit:  # an extra of the default line: the line itself (measured before this) must still come out
                raise EndToEndStalled()
            os._exit(3)
        if any(rcs) or any(r.errors for r in results):
            raise SystemExit("load generator failed: rc %s, %d stream errors" % (rcs, sum(r.errors for r in results)))
        tot = Summed()
        for name in ("access_units", "outputs", "pcm_frames", "pcm_bytes", "errors", "input_full"):
            setattr(tot, name, sum(getattr(r, name) for r in results))
        return tot

    def all_stats():
        acc = {}
        for sc in scheds:
            for k, v in sc.stats().items():
                acc[k] = acc.get(k, 0) + v
        return acc
    if args.warmup:
        run(args.warmup)
    before = all_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    cpu0 = os.times()
    t0 = time.perf_counter()
    res = run(args.steps)
    cpu1 = os.times()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = sharding.reduce_elapsed(time.perf_counter() - t0, device)
    after = all_stats()
    for sc in scheds:
        sc.close()
    for extra in engines[1:]:
        extra.close()
    threads = max(1, threads // n_sched) * n_sched
    if rank != 0:
        return
    st = {k: after[k] - before[k] for k in ("ticks", "frames", "outputs", "parse_ns", "tick_ns", "idle_ns", "deliver_ns")}
    value = world * res.access_units / elapsed
    out = {
        "metric": ("MP3 576-sample granules/s (whole node) + xrealtime, end to end through the batch scheduler" if is_mp3 else
                   "AAC-LC 1024-sample frames/s (whole node) + xrealtime, end to end through the batch scheduler"),
        "value": value, "unit": "granules/s" if is_mp3 else "frames/s", "x_realtime": value / (src_rate / float(unit_frames)), "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "reference fixture %s (%d access units), looped" % (args.clip, units),
        "config": {"workload": "end_to_end: %d %s streams x %d %s, %d Hz %d ch -> %s Hz %s s16, host entropy "
                               "decode on %d threads + GPU ticks" % (args.streams, "MPEG Layer III" if is_mp3 else "ADTS AAC-LC", units * args.steps,
                                                                     "granules" if is_mp3 else "access units", src_rate, src_ch,
                                                                     args.out_rate or src_rate,
                                                                     "mono" if args.out_channels == 1 else "source-channel", threads),
                   "streams_per_gpu": args.streams, "entropy_threads": threads, "feeder_threads": feeders, "host_cores": cores,
                   "host_threads_on_numa_node": os.environ.get("SK_BENCH_PINNED_NODE"),
                   "front_end": ("host threads: MP3 framing, reservoir, scale factors, Huffman codes; i16 lines + granule records over PCIe; "
                                 "k_mp3_requant + k_mp3_hybrid on the device") if is_mp3 else
                                ["host threads (f32 spectra over PCIe)", "gpu (k_aac_entropy_parse/link/finish, one access unit per lane)",
                                 "host Huffman decode, i16 + side records over PCIe, k_aac_expand_q/link/finish on the device"][front_mode],
                   "schedulers": n_sched, "lanes": int(after.get("lanes", n_sched)),
                   "host_cores_busy": ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / elapsed,
                   "host_cores_busy_system": (cpu1.system - cpu0.system) / elapsed,
                   "parallelism": "streams sharded, %d rank(s), no collective" % world},
        "scheduler": {"ticks": st["ticks"], "frames_per_tick": st["frames"] / max(st["ticks"], 1),
                      "entropy_us_per_frame": st["parse_ns"] / max(st["frames"], 1) / 1e3,
                      "entropy_thread_utilisation": st["parse_ns"] / (elapsed * 1e9 * threads),
                      "gpu_tick_ms": st["tick_ns"] / max(st["ticks"], 1) / 1e6,
                      "submission_thread_busy": st["tick_ns"] / (elapsed * 1e9),
                      "delivery_thread_busy": st["deliver_ns"] / (elapsed * 1e9),
                      "submission_thread_waiting": st["idle_ns"] / (elapsed * 1e9),
                      "input_full_events": res.input_full, "pcm_bytes_out": res.pcm_bytes},
        "roofline": None,
        "note": "host-bound: the entropy threads limit this number; the device-resident rooflines are the default workload's",
    }
    # what was decoded: PcmStats of stream 0 (16-bit samples as f32), and the same stream decoded by the CPU chain
    captured = np.frombuffer(cap_buf[:int(cap_len[0])].tobytes(), "<i2")
    out["pcm_stats"] = dict(pcm_stats_s16(lg, captured), stream=0, sample_rate=args.out_rate or src_rate, channels=out_ch,
                            access_units=units * args.steps)
    if args.clip == E2E_CLIP and (args.out_rate or src_rate) == 16000 and out_ch == 1 and not args.no_cpu_baseline:
        want, _ = cpu_chain_decode(clip, args.steps)  # cpu_baseline leg: the checker's decode of the same stream
        cpu = pcm_stats_s16(lg, want)
        same_len = want.size == captured.size
        d = np.abs(want.astype(np.int32) - captured.astype(np.int32)) if same_len else None
        if same_len and os.environ.get("SK_BENCH_DUMP_DIFF") and d.size and int(d.max()) > 1:  # where a run leaves the CPU chain by more than the FIR's LSB
            bad = np.flatnonzero(d > 1)
            sys.stderr.write("pcm diff > 1 LSB at %d samples, first %s .. last %s; per 4096-frame chunk (1365.33 outputs): %s\n"
                             % (bad.size, bad[:12].tolist(), bad[-4:].tolist(), sorted(set((bad * 3 // 4096).tolist()))[:20]))
            for i in bad[:24]:
                sys.stderr.write("   [%d] want %d got %d\n" % (i, int(want[i]), int(captured[i])))
        out["pcm_stats"]["cpu_chain"] = dict(cpu, max_abs_diff_lsb=int(d.max()) if same_len and d.size else None,
                                             differing_fraction=float((d > 0).mean()) if same_len and d.size else None,
                                             same_sample_count=bool(same_len),
                                             note="host front-end + oracle tail on the same bytes; the FIR sums its products in another "
                                                  "order than the CPU chain: +-1 LSB on < 1 % of the samples, so the checksums differ")
    if cpu_baseline:
        out["cpu_baseline"] = cpu_baseline
    if emit:
        print(json.dumps(out))
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command as ranks 0..N-1 (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set),
    relay rank 0's JSON line, fail if any rank fails.  The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if any(codes) or not lines:
        sys.stderr.write("bench.py --gpus %d: rank exit codes %s\n" % (n, codes))
        return next((c for c in codes if c), 1)
    line = json.loads(lines[-1])
    if line.get("n_gpus") != n:
        sys.stderr.write("bench.py: rank 0 reported n_gpus=%r for --gpus %d\n" % (line.get("n_gpus"), n))
        return 1
    print(lines[-1])
    return 0


def gpu_numa_node(local_rank):
    """NUMA node of GPU `local_rank` from sysfs (the amdgpu render nodes in PCI order), or -1: no GPU call is made"""
    try:
        cards = []
        base = "/sys/class/drm"
        for name in sorted(os.listdir(base)):
            if name.startswith("renderD"):
                dev = os.path.realpath(os.path.join(base, name, "device"))
                vendor = open(os.path.join(dev, "vendor")).read().strip()
                if vendor == "0x1002":
                    cards.append(dev)
        cards.sort()
        if local_rank < len(cards):
            return int(open(os.path.join(cards[local_rank], "numa_node")).read().strip())
    except (OSError, ValueError):
        pass
    return -1


def pin_to_gpu_node(torch, device_index):
    """One rank on a two-socket host: keep this process's threads (entropy, submission, delivery, load generator) and the
    pinned buffers they allocate on the GPU's own NUMA node.  The pool's cgroup bounds CPU TIME (16 cores' worth), not
    placement: unpinned, the threads roam over both sockets and whole-decode runs of one build differ by 25 %.  Returns
    the node, or None when the topology cannot be read (then nothing changes).  SK_BENCH_PIN=none switches it off."""
    if os.environ.get("SK_BENCH_PIN") == "none":
        return None
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        mine = [c for c in sorted(os.sched_getaffinity(0)) if cpu_node(c) == node]
        if len(mine) < 4:
            return None
        os.sched_setaffinity(0, mine)
        return node
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def cpu_node(cpu):
    try:
        for name in os.listdir("/sys/devices/system/cpu/cpu%d" % cpu):
            if name.startswith("node") and name[4:].isdigit():
                return int(name[4:])
    except OSError:
        pass
    return 0


def rank_cpu_budget(world, rank, gpu_nodes=None, cpus=None):
    """The host cores of rank `rank` of `world`: the process's affinity set cut into `world` disjoint shares of equal size
    (+-1), each rank's share on its GPU's NUMA node where the node has enough cores left -- so that N ranks on one host do
    not fight over the same cores (every rank runs entropy threads, a submission thread, delivery threads and a load
    generator) and a rank's pinned buffers and threads sit beside its GPU.  Pure function of its arguments (tested on CPU)."""
    if cpus is None:
        cpus = sorted(os.sched_getaffinity(0))[:usable_cores()]  # a cgroup CPU quota below the affinity count bounds the whole job
    cpus = sorted(cpus)
    if world <= 1:
        return cpus
    gpu_nodes = list(gpu_nodes) if gpu_nodes is not None else [gpu_numa_node(r) for r in range(world)]
    by_node = {}
    for c in cpus:
        by_node.setdefault(cpu_node(c), []).append(c)
    want = [len(cpus) // world + (1 if r < len(cpus) % world else 0) for r in range(world)]
    shares = [[] for _ in range(world)]
    order = sorted(range(world), key=lambda r: (gpu_nodes[r], r))
    for r in order:  # first the cores of the rank's own node ...
        pool = by_node.get(gpu_nodes[r], [])
        while pool and len(shares[r]) < want[r]:
            shares[r].append(pool.pop(0))
    left = sorted(c for pool in by_node.values() for c in pool)
    for r in order:  # ... then whatever is left, in order
        while left and len(shares[r]) < want[r]:
            shares[r].append(left.pop(0))
    return sorted(shares[rank])


def print_line_and_leave_if_stalled(out, stalled):
    """the result line always comes out; a run in which any leg stalled then leaves with code 3 -- at once, without tearing the
    engine down (a device that stopped answering would hold the teardown too; the record is on stderr)"""
    print(json.dumps(out))
    if stalled:
        sys.stdout.flush()
        os._exit(3)  # a stall is never rc 0, whichever leg of the line it was in


def dry_run(args, world, rank):
    """SK_BENCH_DRY_RUN=1: the launch / rendezvous / reduction skeleton of the bench on gloo without a GPU (the CPU test
    of `--gpus N`): every rank joins, the timed region is reduced with max, rank 0 prints who it saw."""
    import torch.distributed as dist
    from soundkit_amd import sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    elapsed = sharding.reduce_elapsed(0.001 * (rank + 1))
    ranks = sharding.sum_units(1)
    mine = rank_cpu_budget(world, rank)
    gathered = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"dry_run": True, "n_gpus": world, "ranks_seen": int(ranks), "steps": args.steps, "warmup": args.warmup, "elapsed_max_s": elapsed,
               "cpu_budget": gathered}
        stalled = os.environ.get("SK_BENCH_DRY_STALL") == "1"  # the test of the exit code: as if the end_to_end extra had stalled
        if stalled:
            out["end_to_end"] = {"error": "stalled"}
        print_line_and_leave_if_stalled(out, stalled)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "aac_synth", "fir", "resample", "pcm", "end_to_end"])
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--layout", default="frame", choices=["frame", "stream"],
                    help="packing of the batch: frame-major [frame][stream] (one tick of every stream after another) or stream-major")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-all-cores", action="store_true", help="(default now; kept so that older command lines still run)")
    ap.add_argument("--cpu-baseline-single-core-only", action="store_true",
                    help="skip the one-process-per-core leg of the CPU baseline (saves ~10 s)")
    ap.add_argument("--separate-s16", action="store_true",
                    help="pipeline: f32 FIR output and a separate f32 -> interleaved s16 kernel instead of the conversion in the FIR's "
                         "epilogue (same bytes out, tests/test_pipeline_gpu.py; the chain measured before the bf16 FIR, DESIGN.md 4.2)")
    ap.add_argument("--fused-s16", action="store_true", help="(default now; kept so that older command lines still run)")
    ap.add_argument("--mix", action="store_true",
                    help="aac_synth / pipeline: SURVEY 8d's mixed-sequence batch (10 %% EightShort, each bracketed by LongStart / LongStop) "
                         "instead of OnlyLong; adds `mix` to the JSON: launch time per frame class from three timed batches")
    ap.add_argument("--chain", default="s16", choices=["s16", "f32"],
                    help="pipeline: what crosses HBM between synthesis and FIR -- s16 as in the reference worker (default), or the f32 "
                         "PCM (round 1's chain)")
    ap.add_argument("--tail", default="separate", choices=["separate", "fused"],
                    help="pipeline, s16 chain: the two launches (synthesis to planar s16, then the FIR) or sk_aac_plan_run_tail_s16_dev, "
                         "the same work as one launch with the PCM kept in LDS.  (In a library built with PACKED_F32=1 that kernel is "
                         "withdrawn -- the platform computes it wrongly at this batch size, profiles/r04_lanes_corruption.md -- and "
                         "'fused' runs it through SK_AAC_TAIL_ONE_LAUNCH=1 with \"valid\": false in its line)")
    ap.add_argument("--no-extras", action="store_true",
                    help="pipeline: skip the two extra measurements the default line carries (the mixed-window synthesis launch and a "
                         "few seconds of the whole decode through the scheduler)")
    ap.add_argument("--entropy-threads", type=int, default=0, help="end_to_end: host threads for the AAC front-end (0 = cores - 1, split over ranks)")
    ap.add_argument("--tick-wait-us", type=int, default=0, help="end_to_end: how long a non-empty batch waits for more frames (0 = library default 200)")
    ap.add_argument("--tick-frames", type=int, default=0, help="end_to_end: access units per GPU tick, whole batch (0 = library default)")
    ap.add_argument("--stream-frames-per-tick", type=int, default=0, help="end_to_end: access units one stream may contribute to a tick (0 = library default)")
    ap.add_argument("--schedulers", type=int, default=1,
                    help="end_to_end: independent engine + scheduler pairs on the GPU, each with its share of the streams and threads "
                         "(their ticks overlap on the device)")
    ap.add_argument("--lanes", type=int, default=0, help="end_to_end: engines per scheduler (sk_pipeline_config.lanes; 0 = library default: 2 with --gpu-entropy, else 1)")
    ap.add_argument("--gpu-entropy", action="store_true", help="end_to_end: run the AAC front-end on the GPU too (host threads only frame ADTS)")
    ap.add_argument("--front-end", default=None, choices=["host", "quant", "gpu"],
                    help="end_to_end: where the AAC front-end runs -- host threads (f32 spectra over PCIe), host Huffman decode + device "
                         "dequantisation / PNS / stereo tools / TNS (i16 + side records over PCIe), or all of it on the GPU")
    ap.add_argument("--clip", default="aac-stereo-48k.adts", help="end_to_end: ADTS file under tests/golden/aac to loop")
    ap.add_argument("--feeders", type=int, default=4, help="end_to_end: producer/consumer threads of the load generator (2 + 2 measured best on the 16-core share: profiles/r03_ab_feeders.md)")
    ap.add_argument("--out-rate", type=int, default=16000, help="end_to_end: DecodeOptions.output_sample_rate (0 = source rate)")
    ap.add_argument("--out-channels", type=int, default=1, help="end_to_end: DecodeOptions.output_channels (0 = source)")
    args = ap.parse_args()
    if args.tail == "fused":
        os.environ["SK_AAC_TAIL_ONE_LAUNCH"] = "1"  # only read by a packed-f32 build, whose line is then marked invalid

    # `python bench.py --gpus N` on its own: become the launcher of N ranks (before anything here touches the GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never print a line whose n_gpus is not what was asked for
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("SK_BENCH_DRY_RUN") == "1":
        return dry_run(args, world, rank)
    if world > 1:  # each rank on its own share of the host's cores, beside its GPU (before any thread exists)
        try:
            os.sched_setaffinity(0, rank_cpu_budget(world, local_rank))
        except OSError:
            pass

    # child processes are started before anything here initialises the GPU
    all_cores = whole_decode = None
    if not args.no_cpu_baseline and args.workload in CPU_BASELINES and world == 1 and not args.cpu_baseline_single_core_only:
        all_cores = cpu_baseline_all_cores(args.workload)
    if not args.no_cpu_baseline and args.workload in ("pipeline", "end_to_end") and world == 1 and rank == 0:
        # the whole decode on the host's cores, beside the end_to_end figure (north_star: "the reference ... CPU path timed on the
        # node's own host cores in the same run")
        whole_decode = cpu_baseline_all_cores("decode", 6.0)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import soundkit_amd
    pinned_node = pin_to_gpu_node(torch, local_rank) if world == 1 else None  # several ranks: rank_cpu_budget above already placed them
    os.environ["SK_BENCH_PINNED_NODE"] = "none" if pinned_node is None else str(pinned_node)
    eng = soundkit_amd.Engine(local_rank, max(args.streams, 16))
    ext = torch.cuda.ExternalStream(eng.hip_stream, device=device)

    if args.workload == "end_to_end":
        end_to_end(args, eng, torch, dist, world, rank, device, cpu_baseline=whole_decode)
        if world > 1:
            dist.destroy_process_group()
        eng.close()
        return

    streams, frames, ch = args.streams, args.frames, 2
    kernel_ms = {}

    def timed(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ext)
        fn()
        b.record(ext)
        kernel_ms.setdefault(name, []).append((a, b))

    if args.workload in ("aac_synth", "pipeline"):
        coeffs = seeded_spectra(torch, device, streams, frames, ch, stream0=rank * streams) * SPECTRUM_GAIN  # [stream][frame]
        sids = np.array([eng.open_stream(48000, ch) for _ in range(streams)], np.uint32)
        shape_of_frame = (np.arange(frames) & 1).astype(np.uint8)  # Sine / KBD alternate
        if args.layout == "frame":
            coeffs = coeffs.view(streams, frames, ch, 1024).transpose(0, 1).contiguous().view(-1, ch, 1024)
            ids = np.tile(sids, frames)
            shapes = np.repeat(shape_of_frame, streams)[:, None].repeat(2, 1)
        else:
            ids = np.repeat(sids, frames)
            shapes = np.tile(shape_of_frame, streams)[:, None].repeat(2, 1)
        def sequences(shorts_per_bracket):
            """window sequence of every (stream, frame): OnlyLong, or -- SURVEY 8d's mixed variant -- per ten frames one
            LongStart, `shorts_per_bracket` EightShort, one LongStop (dsp.rs:230-338), each stream shifted by its index"""
            pattern = np.zeros(10, np.uint8)
            if shorts_per_bracket:
                pattern[1] = 1
                pattern[2:2 + shorts_per_bracket] = 2
                pattern[2 + shorts_per_bracket] = 3
            sf = pattern[(np.arange(frames)[None, :] + np.arange(streams)[:, None]) % 10]  # [stream][frame]
            if frames >= 10:  # brackets cut by the batch's edges fall back to OnlyLong (a stream starts and ends long)
                for srow in sf:
                    k = 0
                    while k < frames and srow[k] in (2, 3):
                        srow[k] = 0
                        k += 1
                    k = frames - 1
                    while k >= 0 and srow[k] in (1, 2):
                        srow[k] = 0
                        k -= 1
            flat = sf.T.reshape(-1) if args.layout == "frame" else sf.reshape(-1)
            return np.repeat(flat[:, None], 2, 1)
        seqs = sequences(1 if args.mix else 0)
        descs, n = soundkit_amd.descs_from_arrays(ids, ch, seqs, shapes)
        plan = eng.plan(descs, n)
        assert plan.frames_ok == n
        pcm = torch.empty_like(coeffs)
        units_per_step = streams * frames  # stereo frames
        unit = "frames/s"

        if args.workload == "pipeline":
            if args.layout == "frame":
                stream_stride, frame_stride = ch * 1024, streams * ch * 1024
            else:
                stream_stride, frame_stride = frames * ch * 1024, ch * 1024
            n_out = eng.downsample_out_frames(frames * 1024)
            out_stride = (n_out + 3) // 4 * 4
            fmt_s16 = soundkit_amd.engine.FMT_S16LE
            if args.separate_s16:
                fir_out = torch.empty((streams * ch, out_stride), device=device)
                s16_out = torch.empty((streams, n_out, ch), dtype=torch.int16, device=device)

                def step():
                    timed("k_aac_synth", lambda: plan.run_f32(coeffs, pcm))
                    timed("k_fir_48k_16k", lambda: eng.downsample_48k_16k_frames_dev(pcm, stream_stride, frame_stride, ch,
                                                                                     streams, frames, fir_out, out_stride))
                    timed("k_f32_planar_stereo_to_s16le_batch",
                          lambda: eng.f32_planar_to_bytes_batch_dev(fmt_s16, fir_out, streams, out_stride, n_out, ch, s16_out))
            elif args.chain == "f32":  # f32 PCM between the kernels; the s16 output stage runs in the FIR's epilogue
                s16_stride = (n_out + 7) // 8 * 8
                s16_out = torch.empty((streams, s16_stride, ch), dtype=torch.int16, device=device)

                def step():
                    timed("k_aac_synth", lambda: plan.run_f32(coeffs, pcm))
                    timed("k_fir_48k_16k", lambda: eng.downsample_48k_16k_frames_s16_dev(pcm, stream_stride, frame_stride, ch,
                                                                                         streams, frames, s16_out, s16_stride))
            else:
                # the worker's own data flow (soundkit-decoder lib.rs:1793-1813, 3324-3456): the decoder's output is s16, the
                # resampler is fed s / 32768.  The synthesis kernel writes the s16 itself (planar), the FIR reads it in place.
                s16_stride = (n_out + 7) // 8 * 8
                s16_out = torch.empty((streams, s16_stride, ch), dtype=torch.int16, device=device)
                del pcm
                pcm16 = torch.empty(coeffs.shape, dtype=torch.int16, device=device)

                def separate_step():
                    timed("k_aac_synth", lambda: plan.run_s16_planar(coeffs, pcm16))
                    timed("k_fir_48k_16k", lambda: eng.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, stream_stride, frame_stride, ch,
                                                                                                streams, frames, s16_out, s16_stride))

                def fused_step():
                    timed("k_aac_tail", lambda: plan.run_tail_s16(coeffs, stream_stride, ch, frames, s16_out, s16_stride))
                step = fused_step if args.tail == "fused" else separate_step
            chain_note = {"f32": "f32 PCM between the kernels (not the reference's data flow: it narrows to s16 before resampling)",
                          "s16": "s16 PCM between the kernels as in the reference worker (decode_aac_access_unit -> "
                                 "audio_data_to_f32_channels: s / 32768)"}["f32" if args.separate_s16 else args.chain]
            workload = ("aac_lc decode tail: %d streams x %d frames, 48 kHz stereo: IMDCT+window+OLA -> 48k->16k MFMA FIR -> "
                        "interleaved s16 (%s), %s-major batch; %s" % (streams, frames, "separate kernel" if args.separate_s16 else "in the FIR epilogue", args.layout, chain_note))
        else:
            def step():
                timed("k_aac_synth", lambda: plan.run_f32(coeffs, pcm))
            workload = "aac_lc_synth: %d streams x %d frames, 48 kHz stereo, IMDCT+window+OLA, f32 planar out, %s-major batch" % (streams, frames, args.layout)
    elif args.workload == "pcm":
        # soundkit::audio_bytes / the worker's output stage as one elementwise pass: f32 -> s16 with the reference's
        # float_sample_to_i16 rounding, 512 Mi samples (2 GiB in, 1 GiB out)
        n_samples = 1 << 29
        g = torch.Generator(device=device).manual_seed(SEED0 + rank)
        x = torch.rand(n_samples, generator=g, device=device) * 2.2 - 1.1
        y = torch.empty(n_samples, dtype=torch.int16, device=device)
        units_per_step = n_samples
        unit = "samples/s"

        def step():
            timed("k_convert", lambda: eng.pcm_convert_dev("FLOAT_TO_I16_ROUND", x, y, n_samples))
        workload = "audio_bytes conversion FLOAT_TO_I16_ROUND: %d f32 samples -> s16" % n_samples
    elif args.workload == "resample":
        # the reference's production option (soundkit-decoder lib.rs:4740-4744): 44.1 kHz sources to 16 kHz through the
        # generic-ratio resampler (rubato SincFixedIn, Linear: two 256-tap dot products per output, k_sinc_resample)
        frames_in = 44100
        rows = streams * ch
        g = torch.Generator(device=device).manual_seed(SEED0 + rank)
        x = torch.rand((rows, frames_in), generator=g, device=device) * 2 - 1
        n_out = eng.downsample_out_frames(frames_in, 44100, 16000)
        y = torch.empty((rows, n_out), device=device)
        units_per_step = streams  # stream-seconds
        unit = "stream-seconds/s"

        def step():
            if os.environ.get("SK_BENCH_RESAMPLER_EXACT") == "1":
                eng.set_resampler_exact(True)
            timed("k_sinc_resample", lambda: eng.downsample_dev(x, frames_in, rows, frames_in, 44100, 16000, y, n_out))
        workload = "downsample_audio 44.1k->16k (generic ratio): %d streams x 2 ch x 1 s, f32" % streams
    else:
        frames_in = 48000
        rows = streams * ch
        g = torch.Generator(device=device).manual_seed(SEED0 + rank)
        x = torch.rand((rows, frames_in), generator=g, device=device) * 2 - 1
        n_out = eng.downsample_out_frames(frames_in)
        y = torch.empty((rows, n_out), device=device)
        units_per_step = streams  # stream-seconds
        unit = "stream-seconds/s"

        def step():
            timed("k_fir_48k_16k", lambda: eng.downsample_48k_16k_dev(x, frames_in, rows, frames_in, y, n_out))
        workload = "downsample_audio 48k->16k: %d streams x 2 ch x 1 s, f32" % streams

    torch.cuda.synchronize()  # inputs were produced on torch's stream; the engine runs on its own
    for _ in range(args.warmup):
        step()
    eng.synchronize()
    kernel_ms.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    from soundkit_amd import sharding
    elapsed = sharding.reduce_elapsed(time.perf_counter() - t0, device)  # max over ranks
    synth_f32_ms = None
    if rank == 0 and args.workload == "pipeline" and args.chain == "s16" and not args.separate_s16:
        # reference point for the roofline: the same kernel writing f32 PCM (8 KiB per channel-frame instead of 6), a few
        # launches outside the timed region (profiles/r02_pmc_aac_synth.md has both under the counters).
        scratch = torch.empty_like(coeffs)
        for _ in range(2):
            plan.run_f32(coeffs, scratch)
        eng.synchronize()
        a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(ext)
        for _ in range(5):
            plan.run_f32(coeffs, scratch)
        b0.record(ext)
        eng.synchronize()
        synth_f32_ms = a0.elapsed_time(b0) / 5
        del scratch
    mix_report = None
    if args.mix and args.workload in ("aac_synth", "pipeline") and rank == 0:
        # The synthesis alone on three batches of the same spectra: OnlyLong; one EightShort per bracket; three per bracket.
        # Channels without an EightShort frame run the two-channel long kernel (LongStart / LongStop are the same code with
        # another window table); in the two mixes every channel has EightShort frames, so all of their frames run the general
        # one-channel kernel: two unknowns there (a long-transform frame, a short frame) from the two mixes.
        def synth_ms(shorts):
            d2, n2 = soundkit_amd.descs_from_arrays(ids, ch, sequences(shorts), shapes)
            p2 = eng.plan(d2, n2)
            scratch = torch.empty_like(coeffs)
            for _ in range(3):
                p2.run_f32(coeffs, scratch)
            eng.synchronize()
            a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(ext)
            for _ in range(10):
                p2.run_f32(coeffs, scratch)
            b0.record(ext)
            eng.synchronize()
            p2.destroy()
            seq0 = sequences(shorts)[:, 0]
            return a0.elapsed_time(b0) / 10, int((seq0 == 2).sum()) * ch, int(((seq0 == 1) | (seq0 == 3)).sum()) * ch
        t_long, _, _ = synth_ms(0)
        t1, n_short1, n_trans1 = synth_ms(1)
        t3, n_short3, n_trans3 = synth_ms(3)
        total_cf = streams * frames * ch
        c_pair = t_long / total_cf
        # t = c_general_long * (total - n_short) + c_short * n_short
        A = np.array([[total_cf - n_short1, n_short1], [total_cf - n_short3, n_short3]], np.float64)
        c_glong, c_short = np.linalg.solve(A, np.array([t1, t3]))
        mix_report = {"k_aac_synth_ms": {"only_long": t_long, "one_short_per_bracket (10 % EightShort)": t1, "three_shorts_per_bracket": t3},
                      "channel_frames": total_cf, "short_frames_in_mix": n_short1, "transition_frames_in_mix": n_trans1,
                      "ns_per_channel_frame": {"long-transform frame, two-channel kernel (channels without EightShort)": c_pair * 1e6,
                                               "long-transform frame (OnlyLong / LongStart / LongStop), general kernel": c_glong * 1e6,
                                               "EightShort frame, general kernel": c_short * 1e6},
                      "cost_relative_to_the_general_kernels_long_frame": {"EightShort": c_short / c_glong},
                      "note": "launch time attributed per frame class (chip-wide average, f32 output).  LongStart / LongStop frames "
                              "run the long path with tabulated windows; a wave that meets an EightShort frame runs "
                              "synth_rare_frame for it (aac_synth.hip)"}

    if rank == 0:
        per_kernel = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in kernel_ms.items()}
        value = world * units_per_step * args.steps / elapsed
        out = {
            "metric": "AAC-LC 1024-sample frames/s (whole node) + xrealtime, batch=4096 48 kHz stereo",
            "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (synthesis f32; s16 between the kernels as in the reference; FIR: the 16-bit samples as 2 x f16 exactly, the "
                      "f32 taps as 2 x f16 to 2^-24, f32 accumulate; 1.0e-7 rel. RMS vs f64)" if args.workload == "pipeline" and args.chain == "s16" and not args.separate_s16
                      else "f32 (FIR: exact 3 x bf16 split of both operands, f32 accumulate; 2.2e-7 rel. RMS vs f64)"
                      if args.workload in ("pipeline", "fir") else "f32"),
            "data": "synthetic",
            "config": {"workload": workload, "streams_per_gpu": streams, "frames_per_stream": frames,
                       "sample_rate": 48000, "channels": ch, "seed": "0x12345678 + stream*0x9e3779b9 + frame*2 + ch", "spectrum": "dsp.rs:725-738 seeded_spectrum x %g" % SPECTRUM_GAIN,
                       "parallelism": "streams sharded, %d rank(s), no collective" % world},
        }
        out["config"]["kernel_build"] = ("packed-f32 instructions (make PACKED_F32=1: needs the GPU to itself)" if soundkit_amd._lib.lib.sk_kernels_use_packed_f32()
                                         else "no packed-f32 instructions (the default: immune to the platform's matrix-instruction co-residency defect, "
                                              "profiles/r04_lanes_corruption.md; 2-3 % slower than PACKED_F32=1)")
        if args.tail == "fused" and args.workload == "pipeline" and soundkit_amd._lib.lib.sk_kernels_use_packed_f32():
            out["valid"] = False
            out["invalid_because"] = ("k_aac_tail runs synthesis waves and matrix-instruction waves on the same SIMDs; at this batch the "
                                      "platform computes it wrongly (3 % of the samples, differently in every run): profiles/r04_lanes_corruption.md")
        rl = {}
        if "k_aac_synth" in per_kernel:
            # aac-wasm-bench lib.rs:526-549: 1/rtf summed over the batch -- of the decode TAIL this workload is (no entropy
            # front-end in it); the whole decode's figure is in `end_to_end`
            out["x_realtime_decode_tail"] = value / 46.875
            ms = per_kernel["k_aac_synth"]
            # algorithmic bytes of this variant: 4 KiB in + 4 KiB out per channel-frame, the overlap delay
            # crosses HBM once per channel per launch (in + out); canonical figure charges it every frame
            s16_between = args.workload == "pipeline" and args.chain == "s16" and not args.separate_s16
            out_b = 2048 if s16_between else 4096  # planar s16 or f32 PCM out per channel-frame
            variant_bytes = streams * frames * ch * (4096 + out_b) + streams * ch * 8192
            canonical_bytes = streams * frames * (32768 - (4096 if s16_between else 0))  # minus 2048 B per channel-frame for s16 out
            rl["k_aac_synth"] = {
                "kernel": "k_aac_synth", "bound": "hbm", "achieved": variant_bytes / (ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": variant_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic("aac_synth_s16out" if s16_between else "aac_synth", streams=streams, frames=frames, channels=ch),
                "avg_launch_ms": ms,
                "variant": "delay on-chip: %d B per channel-frame (4096 in + %d out) + 8192 B per channel per launch" % (4096 + out_b, out_b),
                "achieved_canonical_32768B_per_stereo_frame": canonical_bytes / (ms * 1e-3) / 1e9,
            }
            if s16_between:
                rl["k_aac_synth"]["note"] = ("s16-output variant (k_aac_synth_pair<true>: two channels per wave, float_sample_to_i16 fused); the "
                                             "f32-output instance of the same kernel, timed after the run, is in same_kernel_f32_out")
                rl["k_aac_synth"]["kernel"] = "k_aac_synth_pair (channels without a partner: k_aac_synth)"
                if synth_f32_ms:
                    f32_bytes = streams * frames * ch * 8192 + streams * ch * 8192
                    rl["k_aac_synth"]["same_kernel_f32_out"] = {"avg_launch_ms": synth_f32_ms, "achieved": f32_bytes / (synth_f32_ms * 1e-3) / 1e9,
                                                                "frac": f32_bytes / (synth_f32_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                                                                "measured": "5 launches after the timed region, same batch"}
        if "k_aac_tail" in per_kernel:
            out["x_realtime_decode_tail"] = value / 46.875
            ms = per_kernel["k_aac_tail"]
            n_tail_out = eng.downsample_out_frames(frames * 1024)
            # algorithmic bytes of the fused launch: spectra in (4 KiB per channel-frame), 16 kHz interleaved s16 out, the overlap
            # delay once per channel per launch each way -- the s16 PCM between the two stages never leaves the CU
            tail_bytes = streams * frames * ch * 4096 + streams * ch * n_tail_out * 2 + streams * ch * 8192
            rl["k_aac_tail"] = {
                "kernel": "k_aac_tail (synthesis + s16 + 48k->16k MFMA FIR + interleave in one launch)", "bound": "hbm",
                "achieved": tail_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tail_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None, "avg_launch_ms": ms,
                "variant": "4096 B in per channel-frame + 2 B per 16 kHz output sample + 8192 B per channel per launch (delay in/out)",
                "bytes_of_the_two_kernel_chain_it_replaces": streams * frames * ch * 6144 + streams * ch * 8192 + streams * ch * (frames * 1024 * 2 + n_tail_out * 2),
                "matrix_form": "f16, 24 MFMAs per 256 outputs of a channel"}
        if "k_fir_48k_16k" in per_kernel:
            ms = per_kernel["k_fir_48k_16k"]
            fir_in = frames * 1024 if args.workload == "pipeline" else 48000
            n_fir_out = eng.downsample_out_frames(fir_in)
            flops = streams * ch * n_fir_out * 512.0  # 512 flop per output sample
            fused = args.workload == "pipeline" and not args.separate_s16
            pmc_kind = "fir" if args.workload == "fir" else ("fir_pipeline" if args.separate_s16 else
                                                             ("fir_pipeline_s16in" if args.chain == "s16" else "fir_pipeline_s16"))
            # fir_bf16.hip: 24 (s16 rows, f16) to 41 (f32 rows, bf16) MFMAs per 16 x 16 outputs, 1/16 of the time they would take
            # on the f32 matrix path; the bound the line reports is the kernel's own HBM traffic:
            # 4 B per input sample + 4 B (f32) or 2 B (s16) per output sample
            in_b = 2.0 if (fused and args.chain == "s16") else 4.0
            fir_bytes = streams * ch * (fir_in * in_b + n_fir_out * (2.0 if fused else 4.0))
            rl["k_fir_48k_16k"] = {
                "kernel": "k_fir_48k_16k_bf16", "bound": "hbm", "achieved": fir_bytes / (ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fir_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic(pmc_kind, rows=streams * ch, frames=fir_in), "avg_launch_ms": ms,
                "algorithmic_tflops": flops / (ms * 1e-3) / 1e12,
                # issued = algorithmic x MFMAs per tile / 8 (one 16x16x32 MFMA = 256 outputs x 32 taps): f16 form of the s16 rows 24,
                # its bf16 form 36 (SK_FIR_S16_BF16=1), f32 rows 41; f16 and bf16 matrix peaks are the same
                "issued_matrix_tflops": flops * ((3.0 if os.environ.get("SK_FIR_S16_BF16") != "1" else 4.5) if in_b == 2.0 else 5.125) / (ms * 1e-3) / 1e12,
                "matrix_form": ("f16, 24 MFMAs per tile" if os.environ.get("SK_FIR_S16_BF16") != "1" else "bf16, 36 MFMAs per tile") if in_b == 2.0 else "bf16, 41 MFMAs per tile",
                "bf16_mfma_peak_tflops": MFMA_BF16_PEAK_TF,
                "bytes_per_output_sample": 3 * in_b + (2.0 if fused else 4.0)}
        if "k_sinc_resample" in per_kernel:
            ms = per_kernel["k_sinc_resample"]
            n_rs_out = eng.downsample_out_frames(44100, 44100, 16000)
            outs = streams * ch * n_rs_out
            rs_bytes = streams * ch * (44100 * 4.0 + n_rs_out * 4.0)
            # default: the matrix-core form (k_sinc_taps + k_sinc_mfma: the outputs' blended 257-tap filters as the A operand, bf16 x 3
            # planes, six products per 32-sample window); SK_BENCH_RESAMPLER_EXACT=1: the scalar form that keeps rubato's order of
            # operations.  Neither is bound by its 6.7 B of HBM traffic per output; `frac` is against the HBM peak all the same (the
            # schema's choice), the arithmetic figures are beside it.
            exact = os.environ.get("SK_BENCH_RESAMPLER_EXACT") == "1"
            rl["k_sinc_resample"] = {
                "kernel": "k_sinc_resample (scalar form)" if exact else "k_sinc_mfma (+ k_sinc_taps)", "bound": "hbm",
                "achieved": rs_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rs_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms,
                "outputs_per_s": outs / (ms * 1e-3), "algorithmic_tflops": outs * 1026.0 / (ms * 1e-3) / 1e12,
                "vector_f32_peak_tflops_unpacked": 78.6, "bf16_mfma_peak_tflops": MFMA_BF16_PEAK_TF,
                "note": "avg_launch_ms brackets sk_downsample_f32_dev: the host's walk of rubato's f64 time index, two small uploads, the "
                        "launch(es) and a stream synchronisation"}
            out["metric"] = "stream-seconds/s through soundkit::downsample_audio 44.1 kHz -> 16 kHz (generic-ratio sinc resampler)"
        if "k_convert" in per_kernel:
            ms = per_kernel["k_convert"]
            cvt = units_per_step * 6.0  # 4 B in + 2 B out per sample (sk_pcm_op_in_bytes / _out_bytes)
            rl["k_convert"] = {"kernel": "k_convert<SK_PCM_FLOAT_TO_I16_ROUND>", "bound": "hbm", "achieved": cvt / (ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cvt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "avg_launch_ms": ms}
            out["metric"] = "PCM samples/s through soundkit::audio_bytes-style conversion (f32 -> s16, reference rounding)"
            out["config"] = {"workload": workload, "samples_per_gpu": units_per_step,
                             "parallelism": "independent buffers, %d rank(s), no collective" % world}
        if "k_f32_planar_stereo_to_s16le_batch" in per_kernel:
            ms = per_kernel["k_f32_planar_stereo_to_s16le_batch"]
            cvt_bytes = streams * ch * eng.downsample_out_frames(frames * 1024) * 6.0  # 4 B in + 2 B out per sample
            rl["k_f32_planar_stereo_to_s16le_batch"] = {
                "kernel": "k_f32_planar_stereo_to_s16le_batch", "bound": "hbm", "achieved": cvt_bytes / (ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cvt_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None, "avg_launch_ms": ms}
        for entry in rl.values():
            entry["traffic_source"] = ("stored PMC pass of this workload (%s), not a counter read of this run" % os.path.relpath(PMC_SUMMARY, ROOT)
                                       if entry.get("traffic") is not None else None)
        dominant = max(rl, key=lambda k: rl[k]["avg_launch_ms"])
        out["roofline"] = dict(rl[dominant])
        if len(rl) > 1:
            out["roofline"]["note"] = "dominant kernel of the step (largest launch time); every kernel of the chain is in `kernels`"
            out["kernels"] = rl
        if mix_report:
            out["mix"] = mix_report
        stalled = False
        extras = (args.workload == "pipeline" and args.chain == "s16" and not args.separate_s16 and not args.mix and world == 1
                  and not args.no_extras and args.layout == "frame")
        if extras:
            # (a) SURVEY 8d's mixed-window batch through the same synthesis launch (planar s16 out): per ten frames one
            # LongStart, one EightShort, one LongStop, every stream shifted by its index
            def ten_launches(shorts):  # the synthesis launch alone, back to back (inside the timed region it alternates with the FIR)
                d2, n2 = soundkit_amd.descs_from_arrays(ids, ch, sequences(shorts), shapes)
                p2 = eng.plan(d2, n2)
                for _ in range(3):
                    p2.run_s16_planar(coeffs, pcm16)
                eng.synchronize()
                a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(ext)
                for _ in range(10):
                    p2.run_s16_planar(coeffs, pcm16)
                b0.record(ext)
                eng.synchronize()
                p2.destroy()
                return a0.elapsed_time(b0) / 10
            def ten_steps(shorts):  # the same launch where it lives: alternating with the FIR, as in the timed region (per-launch events)
                d2, n2 = soundkit_amd.descs_from_arrays(ids, ch, sequences(shorts), shapes)
                p2 = eng.plan(d2, n2)
                fir = lambda: eng.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, stream_stride, frame_stride, ch, streams, frames, s16_out, s16_stride)
                for _ in range(3):
                    p2.run_s16_planar(coeffs, pcm16)
                    fir()
                eng.synchronize()
                pairs = []
                for _ in range(10):
                    a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a0.record(ext)
                    p2.run_s16_planar(coeffs, pcm16)
                    b0.record(ext)
                    fir()
                    pairs.append((a0, b0))
                eng.synchronize()
                p2.destroy()
                return sum(a0.elapsed_time(b0) for a0, b0 in pairs) / len(pairs)
            long_ms, mixed_ms = ten_launches(0), ten_launches(1)
            mixed_step_ms = ten_steps(1)
            seq0 = sequences(1)[:, 0]
            synth_bytes = streams * frames * ch * 6144 + streams * ch * 8192
            out["mix"] = {"k_aac_synth_ms": {"only_long": long_ms, "mixed": mixed_ms, "only_long_in_the_timed_region": per_kernel["k_aac_synth"],
                                             "mixed_in_the_step": mixed_step_ms},
                          "eight_short_channel_frames": int((seq0 == 2).sum()) * ch, "transition_channel_frames": int(((seq0 == 1) | (seq0 == 3)).sum()) * ch,
                          "channel_frames": streams * frames * ch,
                          "frac_of_hbm_peak_mixed": synth_bytes / (mixed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "frac_of_hbm_peak_mixed_in_the_step": synth_bytes / (mixed_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "note": "planar s16 out; channel pairs whose EightShort frames coincide stay in the two-channel kernel "
                                  "(k_aac_synth_pair<.., true>, synth_rare_pair); only_long / mixed: 10 launches back to back after the timed region "
                                  "(the synthesis alone draws more power than the step and runs ~10 % slower that way: compare only_long with "
                                  "only_long_in_the_timed_region); mixed_in_the_step: the mixed batch's launch alternating with the FIR as the step has it"}
            # (b) the whole decode (ADTS framing, entropy front-end on the GPU, synthesis, 48 -> 16 kHz, mono s16, delivery)
            # through the batch scheduler for a few seconds, in this process: the x-realtime figure of north_star
            del coeffs, pcm16, s16_out
            torch.cuda.empty_cache()
            for sid in sids:
                eng.close_stream(int(sid))
            import copy
            a2 = copy.copy(args)
            a2.workload, a2.front_end, a2.gpu_entropy = "end_to_end", "gpu", True
            a2.steps, a2.warmup = 180, 4   # 4096 streams x 48 units x 180 = 35.4 M access units: 3-4 s at the measured rates (10 M/s)
            try:
                e2e = end_to_end(a2, eng, torch, dist, world, rank, device, emit=False)
            except EndToEndStalled:
                e2e = None
                stalled = True
                out["end_to_end"] = {"error": "stalled: no send and no output for SK_LOADGEN_STALL_S seconds; the scheduler's and the engine's "
                                              "state is on stderr (sk_pipeline_debug_dump)"}
            if e2e is not None:
              out["end_to_end"] = {"value": e2e["value"], "unit": "frames/s", "x_realtime": e2e["x_realtime"], "front_end": e2e["config"]["front_end"],
                                 "host_cores": e2e["config"]["host_cores"], "streams": args.streams, "access_units": args.streams * 48 * a2.steps,
                                 "seconds": e2e["ms_per_step"] * a2.steps / 1000.0, "scheduler": e2e["scheduler"],
                                 "workload": e2e["config"]["workload"], "pcm_stats": e2e.get("pcm_stats"),
                                 "note": "everything in the timed region: ADTS framing on host threads, Huffman decode / stereo tools / TNS, "
                                         "synthesis, streaming 48 -> 16 kHz resampler, mono downmix, s16 pack on the GPU, D2H, delivery; "
                                         "host-fed (the 16-core share frames and delivers), not a kernel figure"}
        if world == 1 and not args.no_cpu_baseline and args.workload in CPU_BASELINES:
            single = CPU_BASELINES[args.workload](6.0 if all_cores else 15.0)
            single["build"] = "-O2 -ffp-contract=off (portable oracle/libsk_oracle.so)"
            # the stated baseline is the host's cores all busy; the single-thread figure sits beside it, as does the
            # reference's own published single-thread rate (whole AAC-LC decode, hardware unstated)
            out["cpu_baseline"] = all_cores or single
            out["cpu_baseline_single_core"] = single
            if whole_decode:
                out["cpu_baseline_whole_decode"] = dict(whole_decode, note="the counterpart of end_to_end.value: everything from ADTS bytes to "
                                                        "16 kHz mono s16 on the host's cores; cpu_baseline above is the decode TAIL only")
            out["cpu_reference_published"] = {"value": 31278.3, "unit": "frames/s", "cores": 1, "kind": "reference-published",
                                              "sample": "soundkit-aac-lc README.md:105, soundkit-lc-reuse: whole AAC-LC decode (entropy + "
                                                        "synthesis), hardware unstated; not measured here"}
        print_line_and_leave_if_stalled(out, stalled)
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
