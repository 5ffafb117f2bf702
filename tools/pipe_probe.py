"""Two workers as in the reference (VideoToDigital thread -> queue -> STC007DataStitcher thread): the binarize stage and the stitch stage of
one tape on two engines, two host threads, two HIP streams, line-record buffers handed over through a queue.  Frames/s of the steady state."""
import os, sys, time, threading, queue
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nbuf = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
H = 486
luma, _ = synth.stc007_frames_torch(n, seed=2, device=dev, noise_sigma=4.0, cyclic=True)
nrec = n * (H + 3)
eb, es = Engine(0), Engine(0)
prio = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s_bin = torch.cuda.Stream(dev, priority=0 if prio >= 0 else -1)
s_st = torch.cuda.Stream(dev, priority=-1 if prio > 0 else 0)
bufs = [torch.empty((nrec + 1, 48), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
stats = torch.empty((n, 32), dtype=torch.uint8, device=dev)
out_p = torch.empty((n * 1470 + 65536, 12), dtype=torch.uint8, device=dev)
out_f = torch.empty((n + 64, 64), dtype=torch.uint8, device=dev)


def run(pipelined):
    eb.reset_stream(); es.reset_stitcher()
    free, ready = queue.Queue(), queue.Queue()
    for b in bufs:
        free.put(b)
    pairs = [0]
    warm = 3

    def binarize():
        fn = 1
        for i in range(batches + warm):
            b = free.get()
            new = i == 0
            eb.binarize_frames(luma, first_frame_no=fn, new_file=new, out_lines=b if new else b[1:], out_stats=stats, stream=s_bin)
            fn += n
            ready.put((b, new))
        ready.put(None)

    def stitch():
        k = 0
        while True:
            item = ready.get()
            if item is None:
                break
            b, new = item
            p, f = es.stitch_frames(b[:1 + nrec] if new else b[1:1 + nrec], out_pairs=out_p, out_frames=out_f, stream=s_st)
            k += 1
            if k == warm:
                torch.cuda.synchronize(dev); t0[0] = time.perf_counter()
            if k > warm:
                pairs[0] += p.shape[0]
            free.put(b)

    t0 = [0.0]
    if pipelined:
        ta, tb = threading.Thread(target=binarize), threading.Thread(target=stitch)
        ta.start(); tb.start(); ta.join(); tb.join()
    else:
        fn = 1
        for i in range(batches + warm):
            b = bufs[0]; new = i == 0
            eb.binarize_frames(luma, first_frame_no=fn, new_file=new, out_lines=b if new else b[1:], out_stats=stats, stream=s_bin)
            fn += n
            p, f = es.stitch_frames(b[:1 + nrec] if new else b[1:1 + nrec], out_pairs=out_p, out_frames=out_f, stream=s_st)
            if i + 1 == warm:
                torch.cuda.synchronize(dev); t0[0] = time.perf_counter()
            if i + 1 > warm:
                pairs[0] += p.shape[0]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0[0]
    return dt / batches * 1e3, pairs[0]


for mode in (False, True, False, True):
    ms, pr = run(mode)
    print(f"{'two threads, two streams' if mode else 'one after the other      '}: {ms:.3f} ms per {n}-frame batch = {n / ms / 1e3:.2f} M frames/s ({pr} pairs)", flush=True)
