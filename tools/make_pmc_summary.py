#!/usr/bin/env python3
"""Assemble profiles/<round>_pmc_summary.json -- the file bench.py's roofline.traffic comes from -- out of the per-run
summaries tools/profile_bench.sh leaves under gpurun_out/profiles_<tag>/ (which are copied to profiles/ as well).
  python3 tools/make_pmc_summary.py r02 r02_s16:aac_synth_s16out,fir_pipeline_s16in r02_f32:aac_synth,fir_pipeline_s16"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
out = {"note": "rocprofv3 --kernel-trace --pmc <counters>, separate passes (FETCH_SIZE | WRITE_SIZE | SQ set | inst set), python3 bench.py "
               "--steps 3 --warmup 1 --no-cpu-baseline [...]; per-launch means.  HBM bytes = 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE counts "
               "64 B per 128-B request on wide coalesced reads: MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024."}
for spec in sys.argv[2:]:
    tag, names = spec.split(":")
    synth_name, fir_name = (names.split(",") + [None])[:2]
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", f))
    pmc = json.load(open(os.path.join(src, tag + "_pmc.json")))
    line = pmc.get("_bench_line_under_profiler", {})
    cfg = line.get("config", {})
    streams, frames, ch = cfg.get("streams_per_gpu"), cfg.get("frames_per_stream"), cfg.get("channels")
    for kernel, name, config in (("k_aac_synth", synth_name, {"streams": streams, "frames": frames, "channels": ch}),
                                 ("k_fir_48k_16k", fir_name, {"rows": (streams or 0) * (ch or 0), "frames": (frames or 0) * 1024})):
        if kernel == "k_aac_synth" and name and not name.endswith("s16out"):
            kernel = "k_aac_synth_f32out"  # tools/summarize_pmc.py keys the two output types apart
        if name and kernel in pmc:
            e = pmc[kernel]
            out[name] = {"config": config, "counters": e["counters"], "meta": e.get("meta"), "avg_ns_profiled": e.get("avg_ns_profiled"),
                         "hbm_read_bytes": e.get("hbm_read_bytes"), "hbm_write_bytes": e.get("hbm_write_bytes"),
                         "traffic_bytes": e.get("traffic_bytes"), "source": "profiles/%s_pmc.json" % tag}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print({k: (v.get("traffic_bytes"), v.get("config")) for k, v in out.items() if isinstance(v, dict)})
