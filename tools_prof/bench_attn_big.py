import sys
sys.argv = ["x"]
exec(open("tools_prof/bench_attn.py").read().split("run(3, 128, 384, 3, 1)")[0])
run(4096, 128, 384, 3, check=False)
