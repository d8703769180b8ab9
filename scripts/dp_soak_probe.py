"""Soak of the data-parallel scorer on a ONE-rank RCCL group (what bench.py --force-dp times), graph replay on: N steps of score_clips_dp with rotating inputs - every repetition
of an input set bit-identical to its first occurrence and to the plain forward, device memory flat.  python scripts/dp_soak_probe.py [steps = 400]  (MI355X)"""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import aigv_assessor_amd as pkg
from aigv_assessor_amd import dist_utils, synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dist_utils.force_single_rank_collectives = True
cfg = pkg.internvl2_8b()
B, T = 4, 8
dev = torch.device("cuda", 0)
N = synth.canonical_len(cfg, T)
model = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * N)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
model.eval()
model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=0))
toks = synth.canonical_tokens(cfg, B, T, seed=0)
model.img_context_token_id = toks["img_context_token_id"]
flags = torch.ones(B * T, 1, dtype=torch.long)
pvs = [synth.synthetic_frames(B * T, 448, seed=s).to(dev) for s in range(3)]
plain = [model(mos=None, pixel_values=p, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"]) for p in pvs]
plain = [(o["score1"].clone(), o["logit"].clone()) for o in plain]
model.enable_graph_replay(True)
bad, marks = 0, []
for i in range(steps):
    o = dist_utils.score_clips_dp(model, pvs[i % 3], toks["input_ids"], toks["attention_mask"], flags, toks["labels"], None)
    if i % 50 == 49 or i < 3:
        torch.cuda.synchronize()
        if not (torch.equal(o["score1"], plain[i % 3][0]) and torch.equal(o["logit"], plain[i % 3][1])):
            bad += 1
        free, total = torch.cuda.mem_get_info(dev)
        marks.append((i + 1, (total - free) / 2 ** 20))
torch.cuda.synchronize()
print(f"{steps} steps of score_clips_dp on a one-rank RCCL group (graph replay: {sum(isinstance(v, tuple) for v in model._graphs.values())} captured graphs); checked steps that differ from the plain forward: {bad}")
print("device memory in use (MiB) at steps:", ", ".join(f"{n}: {m:.0f}" for n, m in marks))
assert bad == 0 and marks[-1][1] - marks[3][1] < 64
dist.barrier(); dist.destroy_process_group()
print("DP_SOAK_OK")
