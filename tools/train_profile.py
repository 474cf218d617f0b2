"""Where a whole-model training step spends its GPU time, by autograd operator and operand shape (torch.profiler): the table that
says which eager elementwise / copy / fill launches around the hand-written kernels are worth a kernel of their own.

    python tools/train_profile.py [--workload c3] [--steps 3] [--top 70] > profiles/rNN_train_profile_c3.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from allegro_amd.nn import HipAllegroModel, PreparedGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--stacks", default="", help="comma-separated operator names (e.g. aten::copy_,aten::add_): their device time by the autograd node they ran under")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g, cfg = bench.make_workload(args.workload)
    dtype = {"float32": torch.float32, "float64": torch.float64}[cfg["model_dtype"]]
    torch.manual_seed(0)
    model = HipAllegroModel(**cfg).to(dev).train()
    N = g.num_atoms
    sv = g.shift_vec()
    pos = torch.tensor(g.pos, dtype=dtype, device=dev)
    graph = PreparedGraph(torch.tensor(g.edge_index, device=dev), torch.tensor(g.types, device=dev), N,
                          torch.tensor(sv, dtype=dtype, device=dev) if sv is not None else None)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    f_target = 0.1 * torch.randn(N, 3, dtype=dtype, device=dev)
    ev = model._training_evaluator()

    def step():
        opt.zero_grad(set_to_none=True)
        out = ev.forward({"pos": pos}, graph)
        loss = (out["forces"] - f_target).square().mean() + 1e-3 * (out["total_energy"] / N).square().sum()
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    if args.stacks:
        # per operator and shape: device time by the autograd node (or forward-pass Python op) it ran under
        want = set(args.stacks.split(","))
        agg = {}
        for e in prof.events():
            if e.name not in want:
                continue
            t = getattr(e, "device_time_total", None)
            if t is None:
                t = getattr(e, "cuda_time_total", 0.0)
            if t <= 0:
                continue
            par, node = e.cpu_parent, "(forward)"
            while par is not None:
                if par.name.startswith("autograd::engine::evaluate_function: "):
                    node = par.name.split(": ", 1)[1]
                    break
                par = par.cpu_parent
            key = (e.name, str(e.input_shapes)[:48], node)
            ms, cnt = agg.get(key, (0.0, 0))
            agg[key] = (ms + t / args.steps / 1e3, cnt + 1.0 / args.steps)
        for (name, shapes, node), (ms, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:args.top]:
            print(f"{ms:8.3f} ms {cnt:5.1f}x  {name:12s} {shapes:48s} {node}")
        return
    rows = []
    for ev_ in prof.key_averages(group_by_input_shape=True):
        t = getattr(ev_, "self_device_time_total", None)
        if t is None:
            t = getattr(ev_, "self_cuda_time_total", 0.0)
        if t > 0:
            rows.append((t / args.steps / 1e3, ev_.count / args.steps, ev_.key, str(ev_.input_shapes)[:150]))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"# {args.workload}: self device time per training step by operator and input shapes; total {total:.2f} ms/step")
    for ms, cnt, key, shapes in rows[:args.top]:
        print(f"{ms:8.3f} ms {cnt:6.1f}x  {key[:60]:60s} {shapes}")


if __name__ == "__main__":
    main()
