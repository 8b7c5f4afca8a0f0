"""Timing prototype (not a product path): the default chain of bench.py -- synthesis to planar s16, then the 48k->16k FIR --
cut into time slices, the FIR of slice k on a second HIP stream beside the synthesis of slice k + 1.  The FIR calls here
are one-shot per slice (zero history at the slice's start), so the VALUES at slice edges are not the chain's; the work and
the bytes are.  Question answered: does running the two kernels side by side buy anything on this part?

    python tools/chain_overlap.py [--streams 4096] [--frames 64]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import soundkit_amd
    device = torch.device("cuda:0")
    streams, frames, ch = args.streams, args.frames, 2
    eng = soundkit_amd.Engine(0, streams)
    eng2 = soundkit_amd.Engine(0, 16)   # its stream carries the FIR launches
    s1 = torch.cuda.ExternalStream(eng.hip_stream, device=device)
    s2 = torch.cuda.ExternalStream(eng2.hip_stream, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    coeffs = (torch.rand((frames, streams, ch, 1024), generator=g, device=device) * 2 - 1) * 2500.0   # frame-major
    pcm16 = torch.empty(coeffs.shape, dtype=torch.int16, device=device)
    sids = np.array([eng.open_stream(48000, ch) for _ in range(streams)], np.uint32)
    stream_stride, frame_stride = ch * 1024, streams * ch * 1024
    torch.cuda.synchronize()
    results = {}
    for n_slices in (1, 2, 4, 8, 16):
        per = frames // n_slices
        plans, outs = [], []
        for k in range(n_slices):
            ids = np.tile(sids, per)
            seqs = np.zeros((per * streams, 2), np.uint8)
            shapes = np.repeat(((np.arange(per) + k * per) & 1).astype(np.uint8), streams)[:, None].repeat(2, 1)
            descs, n = soundkit_amd.descs_from_arrays(ids, ch, seqs, shapes)
            plans.append(eng.plan(descs, n))
            n_out = eng.downsample_out_frames(per * 1024)
            stride = (n_out + 7) // 8 * 8
            outs.append((torch.empty((streams, stride, ch), dtype=torch.int16, device=device), stride))
        for overlap in (False, True):
            fir_eng = eng2 if overlap else eng

            def step():
                for k in range(n_slices):
                    c = coeffs[k * per:(k + 1) * per]
                    p = pcm16[k * per:(k + 1) * per]
                    plans[k].run_s16_planar(c, p)
                    if overlap:
                        ev = torch.cuda.Event()
                        ev.record(s1)
                        s2.wait_event(ev)
                    fir_eng.downsample_48k_16k_frames_s16_to_s16_dev(p, stream_stride, frame_stride, ch, streams, per, outs[k][0], outs[k][1])
                if overlap:   # the next step's synthesis overwrites pcm16: it waits for this step's last FIR
                    ev = torch.cuda.Event()
                    ev.record(s2)
                    s1.wait_event(ev)
            for _ in range(3):
                step()
            eng.synchronize(), eng2.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            eng.synchronize(), eng2.synchronize()
            ms = (time.perf_counter() - t0) * 1000 / args.steps
            results[(n_slices, overlap)] = ms
            print("slices %2d  %s  %.3f ms per step" % (n_slices, "two streams" if overlap else "one stream ", ms), flush=True)
        for p in plans:
            p.destroy()
    eng2.close()
    eng.close()


if __name__ == "__main__":
    main()
