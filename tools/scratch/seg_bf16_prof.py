import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
r = bench.seg_eval_bench(torch.device("cuda", 0), 1, "bf16", crf=False, steps=3)
print(json.dumps({k: r[k] for k in ("windows_per_s", "tflops_per_gpu")}))
