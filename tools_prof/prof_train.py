import sys, os, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
r = bench.train_bench(dev, 0, 1, 20, 5)
print(r)
