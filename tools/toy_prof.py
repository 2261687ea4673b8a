import sys, torch, json
sys.path.insert(0, ".")
from demiss_vae_amd.bench_extra import demiss_toy_step
print(json.dumps(demiss_toy_step(torch.device("cuda"), B=int(sys.argv[1]) if len(sys.argv) > 1 else 4096)))
