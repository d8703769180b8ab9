"""Randomised run of the multi-rank scorer on ONE card over gloo (the rehearsal transport of tests/test_gpu_dist.py; RCCL refuses duplicate devices): `world` rank processes, all
with the same seed, draw the same sequence of calls - B = 1..6 clips of T = 8 / 12 / 16 frames, the native SlowFast branch inside or a given motion feature, prefer_gathered or
not, graph replay toggled now and then, ragged clip / frame splits (B not a multiple of world, B < world) - and every rank's `score_clips_dp` result must equal its own plain
one-process `forward` bit for bit.  The long-lived model keeps its captured graphs, grown contexts and cached SlowFast handles from call to call.

    python tests/manual/fuzz_dp.py [world = 2] [n_calls = 40] [seed = 0]          # parent: spawns the ranks;  ... --rank R is a child"""
import os
import random
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(n_calls, seed):
    import torch
    import torch.distributed as dist
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import dist_utils, synth
    from aigv_assessor_amd.modeling import InternVLChatModel
    from aigv_assessor_amd.slowfast import SlowFastR50
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["LOCAL_RANK"] = "0"
    dist_utils.init_dist("pytorch", backend="gloo")
    dev = torch.device("cuda", 0)
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    sd = synth.make_state_dict(cfg, seed=7, rich=True)
    sf_sd = synth.slowfast_state_dict(seed=3)

    def make():
        m = InternVLChatModel(cfg, max_clips=1)
        m.load_state_dict(sd)
        m.eval().cuda()
        m.slowfast_model = SlowFastR50(sf_sd)
        return m
    model, ref = make(), make()
    rng = random.Random(seed)
    graph = False
    for it in range(n_calls):
        if rng.random() < 0.2:
            graph = not graph
            model.enable_graph_replay(graph)
        B, T, branch, prefer, s = rng.randint(1, 6), rng.choice([8, 8, 12, 16]), rng.random() < 0.6, rng.random() < 0.4, rng.randint(0, 3)
        toks = synth.canonical_tokens(cfg, B, T, seed=s)
        model.img_context_token_id = ref.img_context_token_id = toks["img_context_token_id"]
        pv = synth.synthetic_frames(B * T, 224, seed=s).to(dev)
        motion = None if branch else synth.synthetic_motion(B, cfg.motion_dim, seed=s).to(dev)
        flags = torch.ones(B * T, 1, dtype=torch.long)
        plain = ref(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"], motion_feature=motion)
        for rep in range(3 if graph else 1):
            dp = dist_utils.score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, prefer_gathered=prefer)
            torch.cuda.synchronize()
            ok = torch.equal(dp["score1"], plain["score1"]) and torch.equal(dp["logit"], plain["logit"]) and torch.equal(dp["label"], plain["label"])
            assert ok, (rank, it, rep, B, T, branch, prefer, graph)
        if rank == 0 and it % 10 == 9:
            print(f"call {it + 1}/{n_calls} ok", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print(f"FUZZ_DP_OK rank={rank}/{world} calls={n_calls}", flush=True)


def main():
    if "--rank" in sys.argv:
        a = [x for x in sys.argv[1:] if not x.startswith("--")]
        return child(int(a[0]), int(a[1]))
    a = sys.argv[1:]
    world, n_calls, seed = int(a[0]) if a else 2, int(a[1]) if len(a) > 1 else 40, int(a[2]) if len(a) > 2 else 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), str(n_calls), str(seed), "--rank"], env=env, cwd=ROOT))
    rcs = [p.wait(timeout=900) for p in procs]
    print("ranks exited with", rcs)
    raise SystemExit(max(abs(r) for r in rcs))


if __name__ == "__main__":
    main()
