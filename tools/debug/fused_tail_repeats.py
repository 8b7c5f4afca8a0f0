"""The fused decode tail (k_aac_tail: synthesis waves and matrix-instruction FIR work in ONE launch, so both kinds share CUs) at
the headline batch size, run again and again on a fixed input and compared bit for bit with the two-launch chain's result.
python tools/debug/fused_tail_repeats.py [streams] [frames] [runs]"""
import os
import sys

import numpy as np
import torch

os.environ["SK_AAC_TAIL_ONE_LAUNCH"] = "1"  # the entry point is withdrawn; this is what the switch is for
sys.path.insert(0, ".")
import soundkit_amd  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ch = 2
dev = torch.device("cuda:0")
eng = soundkit_amd.Engine(0, streams + 8)
sids = np.array([eng.open_stream(48000, ch) for _ in range(streams)], np.uint32)
g = torch.Generator(device="cpu").manual_seed(5)
coeffs = ((torch.rand((streams * frames, ch, 1024), generator=g) * 2 - 1) * 2.5e5).to(dev)
ids = np.repeat(sids, frames)
seqs = np.zeros((streams * frames, 2), np.uint8)
shapes = np.tile((np.arange(frames) & 1).astype(np.uint8), streams)[:, None].repeat(2, 1)
descs, n = soundkit_amd.descs_from_arrays(ids, ch, seqs, shapes)
plan = eng.plan(descs, n)
stream_stride, frame_stride = frames * ch * 1024, ch * 1024
n_out = eng.downsample_out_frames(frames * 1024)
stride = (n_out + 7) // 8 * 8
ext = torch.cuda.ExternalStream(eng.hip_stream, device=dev)


def reset():
    for sid in sids:
        eng.reset_stream(int(sid))


with torch.cuda.stream(ext):
    pcm16 = torch.zeros(coeffs.shape, dtype=torch.int16, device=dev)
    want = torch.zeros((streams, stride, ch), dtype=torch.int16, device=dev)
    reset()
    plan.run_s16_planar(coeffs, pcm16)
    assert eng.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, stream_stride, frame_stride, ch, streams, frames, want, stride) == n_out
    eng.synchronize()
    print("two launches: rms of the s16 result", float(want.float().pow(2).mean().sqrt()), flush=True)
    again = torch.zeros_like(want)
    bad2 = 0
    for r in range(runs):
        reset()
        again.zero_()
        plan.run_s16_planar(coeffs, pcm16)
        eng.downsample_48k_16k_frames_s16_to_s16_dev(pcm16, stream_stride, frame_stride, ch, streams, frames, again, stride)
        eng.synchronize()
        bad2 += int(not torch.equal(again, want))
    print("two launches repeated:", bad2, "of", runs, "runs differ", flush=True)
    bad, words, worst, unlike_first = 0, 0, 0, 0
    first_fused = None
    for r in range(runs):
        reset()
        again.zero_()
        assert plan.run_tail_s16(coeffs, stream_stride, ch, frames, again, stride) == n_out
        eng.synchronize()
        if first_fused is None:
            first_fused = again.clone()
        unlike_first += int(not torch.equal(again, first_fused))
        ne = again != want
        k = int(ne.sum().item())
        if k:
            bad += 1
            words += k
            worst = max(worst, int((again.int() - want.int()).abs().max().item()))
    print("fused tail:", bad, "of", runs, "runs differ from the two launches,", words, "samples, largest difference", worst, "LSB;", unlike_first, "runs differ from the first fused run", flush=True)
    if bad:
        where = ne.nonzero()
        st = torch.unique(where[:, 0])
        print("last run: streams hit", st.numel(), "of", streams, "first", where[0].tolist(), "output positions of stream", int(st[0]), ":",
              torch.unique(where[where[:, 0] == st[0]][:, 1])[:12].tolist(), flush=True)
