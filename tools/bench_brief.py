"""Prints the figures of a bench.py JSON line a human wants to see first (tools/gpu_round.sh)."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
r = d.get("roofline", {})
print(f"{d['config']['workload'][:40]}: {d['ms_per_step']:.3f} ms/step, {d['value']:.4g} {d['unit']}, roofline {r.get('kernel')} "
      f"{r.get('achieved', 0):.1f}/{r.get('peak', 0)} {r.get('unit')} = {r.get('frac', 0):.3f}")
p = d.get("parity_sample")
if p:
    print(f"  parity ok={p['ok']} max_dF={p['max_dF']:.2e} max_dE={p['max_dE']:.2e} vs_fp64={p.get('vs_fp64', {}).get('hip')} / oracle32 {p.get('vs_fp64', {}).get('oracle_fp32')}")
for k, v in (d.get("stage_ms") or {}).items():
    print(f"  {k:32s} {v['ms']:8.4f} ms x{v['launches']}  {v['GBps']:8.1f} GB/s")
for k, v in (d.get("secondary") or {}).items():
    rr = v["roofline"]
    print(f"  secondary {k}: {v['ms_per_step']:.3f} ms/step, {rr['kernel'][:40]} frac {rr['frac']:.3f}, parity ok={v['parity_sample']['ok']} "
          f"max_dF={v['parity_sample']['max_dF']:.2e}")
