import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
mot = bench._load("mot_amd", os.path.join(bench.PKG_DIR, "__init__.py"))
sdev = bench._load("mot_amd.synth_dev", os.path.join(bench.ROOT, "tools", "synth", "synth_dev.py"))
seqmod = bench._load("mot_amd.sequence", os.path.join(bench.PKG_DIR, "sequence.py"))
F, N = 154, 120000
stride = ((N + 2047) // 2048) * 2048
v, yaw = sdev.load_ego(F)
r = sdev.SequenceRenderer("cuda")
seq, n_seq, _o, _p = r.render([0], F, N, stride, v, yaw)
print(json.dumps(bench.stage_wise_host_buffers(mot, 0, seq, n_seq, stride, v, yaw, seqmod)))
