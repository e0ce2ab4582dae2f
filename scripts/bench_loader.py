"""Input-pipeline figures on the GPU box: device preprocessing time per batch, host read rate, loader-only batches/s, and the
host-side (oracle = scipy, what the reference runs per slice) cost beside them.   python scripts/bench_loader.py [B]"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transception_amd import data as D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="synapse_")
t0 = time.perf_counter()
D.write_synthetic_synapse(tmp + "/train_npz", tmp + "/lists", n_cases=4, slices_per_case=32, size=512, seed=1234)
print(f"wrote 128 synthetic slices in {time.perf_counter() - t0:.1f} s")
ds = D.SynapseSlices(tmp + "/train_npz", tmp + "/lists")

t0 = time.perf_counter()
items = [ds[i] for i in range(64)]
t_read = (time.perf_counter() - t0) / 64
print(f"host read: {t_read * 1e3:.2f} ms/slice on one thread ({1 / t_read:.0f} slices/s)")

img = torch.from_numpy(np.stack([it[0] for it in items[:B]])).to(dev)
lab = torch.from_numpy(np.stack([it[1] for it in items[:B]])).to(dev)
sampler = D.AugmentSampler(1)
for name, augs in (("no augmentation", None), ("sampled augmentation", [sampler.sample(512, 512) for _ in range(B)]),
                   ("worst case (warp+piecewise+blur+noise on every slice)",
                    [D.SliceAugmentation(m=D.affine_rotate_xy(20, 512, 512), disp=np.ones((4, 4, 2), np.float32), shape=(512, 512), blur=True, alpha=1.1,
                                         noise_sigma=1.0, noise_seed=i) for i in range(B)])):
    rec, nr = None, None
    if augs is not None:                                         # chains of imgaug stages: one record array per round
        packed, nr = D.pack_rounds(augs)
        rec = torch.from_numpy(packed).to(dev)
    scratch = {}
    for _ in range(3):
        D.preprocess_batch(img, lab, None, 224, records=rec, scratch=scratch, rounds=nr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        D.preprocess_batch(img, lab, None, 224, records=rec, scratch=scratch, rounds=nr)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"device preprocess, B={B}, 512->224, {name}: {ms:.3f} ms/batch = {B / ms * 1e3:.0f} slices/s")
    if rec is not None:                                          # what a captured step launches: all MAX_ROUNDS rounds, inside a replayed graph
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            D.preprocess_batch(img, lab, None, 224, records=rec, scratch=scratch, rounds=D.MAX_ROUNDS)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"    all {D.MAX_ROUNDS} rounds in a replayed graph: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us/batch")

for aug in (False, True):
    loader = D.DeviceLoader(ds, batch_size=B, img_size=224, device=dev, augment=aug, epochs=4)
    n = 0
    t0 = None
    for x, y in loader:
        if n == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"loader only (augment={aug}): {(n - 1) * B / dt:.0f} slices/s over {n - 1} batches")

from oracle import data_oracle as O
sampler = D.AugmentSampler(2)
t0 = time.perf_counter()
for i in range(8):
    O.preprocess_slice(items[i][0], items[i][1], sampler.sample(512, 512).as_dict(), 224)
print(f"host oracle (scipy) augment+zoom+normalise: {(time.perf_counter() - t0) / 8 * 1e3:.1f} ms/slice/core")
t0 = time.perf_counter()
for i in range(8):
    O.resize_normalize(items[i][0], items[i][1].astype(np.float32), 224)
print(f"host scipy zoom order 3 + order 0 + normalise alone: {(time.perf_counter() - t0) / 8 * 1e3:.1f} ms/slice/core")
