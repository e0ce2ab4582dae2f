"""Plain forward GEMMs at the step's shapes: time, TFLOP/s and algorithmic GB/s (A + B read once, C written once)."""
import sys; sys.argv = sys.argv[:1]
exec(open("/root/repo/scripts/bench_gemm.py").read().split("run(12544, 256, 64)")[0].replace('print(f"M=', 'gb = (M * K + N * K + M * N) * 2.0 * nb / us / 1e3; print(f"{gb:7.0f} GB/s  M='))
for (M, N, K, nb) in ((50176, 256, 64, 3), (50176, 64, 256, 3), (12544, 512, 128, 3), (12544, 128, 512, 3), (3136, 1280, 320, 3), (3136, 320, 1280, 3), (50176, 64, 64, 1), (50176, 256, 64, 1), (50176, 64, 256, 1)):
    run(M, N, K, nb=nb)
