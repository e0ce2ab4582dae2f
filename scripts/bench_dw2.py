import sys; sys.argv = sys.argv[:1]
exec(open("/root/repo/scripts/bench_dw.py").read().split("run(16, 56, 56, 256, 3)")[0])
run(16, 56, 56, 256, 3); run(16, 28, 28, 512, 3); run(16, 56, 56, 64, 3, 3); run(16, 56, 56, 24, 7, 3)
