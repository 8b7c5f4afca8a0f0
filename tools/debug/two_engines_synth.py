"""Two engines on one device, each running its own synthesis launches from its own thread on a fixed input: does a launch ever
produce other bytes than the first one did?  (python tools/debug/two_engines_synth.py [threads] [iterations] [streams] [frames])"""
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, ".")
import soundkit_amd  # noqa: E402

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 16
kind = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda:0")
bad = [0] * n_threads


def work(t):
    eng = soundkit_amd.Engine(0, streams + 8)
    sids = np.array([eng.open_stream(48000, 2) for _ in range(streams)], np.uint32)
    g = torch.Generator(device="cpu").manual_seed(7)
    coeffs = (torch.rand((streams * frames, 2, 1024), generator=g) * 24 - 12).to(dev)
    ids = np.repeat(sids, frames)
    seqs = np.zeros((streams * frames, 2), np.uint8)
    shapes = np.tile((np.arange(frames) & 1).astype(np.uint8), streams)[:, None].repeat(2, 1)
    descs, n = soundkit_amd.descs_from_arrays(ids, 2, seqs, shapes)
    plan = eng.plan(descs, n)
    ext = torch.cuda.ExternalStream(eng.hip_stream, device=dev)
    first = None
    with torch.cuda.stream(ext):
        for it in range(iters):
            for sid in sids:
                eng.reset_stream(int(sid))
            if kind == "f32":
                pcm = torch.empty_like(coeffs)
                plan.run_f32(coeffs, pcm)
            else:
                pcm = torch.empty(coeffs.shape, dtype=torch.int16, device=dev)
                plan.run_s16_planar(coeffs, pcm)
            eng.synchronize()
            h = int(pcm.view(torch.int32).to(torch.int64).sum().item()) if kind == "f32" else int(pcm.to(torch.int64).sum().item())
            if first is None:
                first = (h, pcm.clone())
            elif h != first[0]:
                bad[t] += 1
                if bad[t] <= 3:
                    a, b = pcm.to(torch.float64), first[1].to(torch.float64)
                    diff = (a - b).abs()
                    where = diff.nonzero()
                    rel = diff / b.abs().clamp_min(1e-30)
                    frames_hit = torch.unique(where[:, 0]).numel()
                    print("thread", t, "iteration", it, "differs at", where.shape[0], "elements in", frames_hit, "frames; max abs", float(diff.max()),
                          "max rel", float(rel[diff > 0].max()), "median rel", float(rel[diff > 0].median()), "| reference rms", float(b.pow(2).mean().sqrt()),
                          "first", where[0].tolist(), flush=True)
                    if bad[t] == 1 and len(sys.argv) > 6:  # keep the first hit frames and the frame behind each for a look at the error's shape
                        rows = torch.unique(where[:, 0])[:24].tolist()
                        keep = sorted(set(rows) | {r + 1 for r in rows if (r + 1) % frames})
                        np.savez(sys.argv[6], rows=np.array(keep), hit=np.array(rows), frames=frames, got=pcm[keep].cpu().numpy(), ref=first[1][keep].cpu().numpy(),
                                 coeffs=coeffs[keep].cpu().numpy(), all_hit=torch.unique(where[:, 0]).cpu().numpy())
    plan.destroy()


threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
for th in threads:
    th.start()
for th in threads:
    th.join()
print("threads", n_threads, "iterations", iters, "kind", kind, "deviating launches per thread:", bad)
