"""Generates allegro_amd/csrc/aa_cg_gen.h: straight-line Clebsch-Gordan contraction code for the
tensor-product signatures of standard Allegro layer stacks (l_max 1..3, 1..3 layers, parity=True).

For every signature (irreps of in1 x SH env -> out, enumerated as allegro/nn/_allegro.py:101-183 and
allegro/nn/_strided/_contract.py:53-119 do) three fully unrolled device functions are emitted with
all indices and coefficients as compile-time constants, so operands live in registers:
    fwd : out[k] = sum_p w[p] * sum_nz c * x1[i] * x2[j]
    bx1 : g1[i]  = sum_p w[p] * sum_nz c * go[k] * x2[j]
    bx2 : g2[j]  = sum_p w[p] * sum_nz c * go[k] * x1[i]
plus the non-zero list itself, which the host uses to check that a model's `w3j` buffer
(_contract.py:168) equals the compiled table before selecting the specialised kernels.

Run by allegro_amd/build.py when the header is missing or older than this script.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def signatures():
    from allegro_amd import o3
    from allegro_amd.nn import allegro_layer_irreps, build_w3j

    sigs = {}
    for lmax in (1, 2, 3):
        env = o3.Irreps.spherical_harmonics(lmax)
        for L in (1, 2, 3):
            t = allegro_layer_irreps(lmax, True, L)
            for l in range(L):
                key = (lmax, repr(t[l]), repr(t[l + 1]))
                if key in sigs:
                    continue
                w3j, instr, diag, dims = build_w3j(t[l], env, t[l + 1])
                P = len(instr)
                w = w3j if P > 1 else w3j[None]
                if diag:
                    p, i, k = np.nonzero(w)
                    j = i
                    v = w[p, i, k]
                else:
                    p, i, j, k = np.nonzero(w)
                    v = w[p, i, j, k]
                sigs[key] = dict(lmax=lmax, dims=dims, P=P, diag=diag, nz=list(zip(i.tolist(), j.tolist(), k.tolist(), p.tolist(), v.tolist())),
                                 name=f"{t[l]} x SH{lmax} -> {t[l + 1]}")
    return list(sigs.values())


def emit_fn(name, a_name, b_name, out_name, n_out, entries):
    """entries: list of (out, path, a, b, val).  Emits a 4-type template (operands / weights / output may be
    scalars or 2-vectors: packed two-edge evaluation) plus the single-type convenience overload."""
    lines = [f"  template <typename TO, typename TA, typename TB, typename TW> static __device__ __forceinline__ void "
             f"{name}4(const TA* __restrict__ {a_name}, const TB* __restrict__ {b_name}, const TW* __restrict__ w, "
             f"TO* __restrict__ {out_name}) {{"]
    by_out = {}
    for (o, p, a, b, v) in entries:
        by_out.setdefault(o, {}).setdefault(p, []).append((a, b, v))
    for o in range(n_out):
        if o not in by_out:
            lines.append(f"    {out_name}[{o}] = TO(0);")
            continue
        terms = []
        for p, ents in sorted(by_out[o].items()):
            s = " + ".join(f"TO({v!r}) * {a_name}[{a}] * {b_name}[{b}]" for (a, b, v) in sorted(ents))
            terms.append(f"w[{p}] * ({s})")
        lines.append(f"    {out_name}[{o}] = " + "\n        + ".join(terms) + ";")
    lines.append("  }")
    lines.append(f"  template <typename T> static __device__ __forceinline__ void {name}(const T* __restrict__ {a_name}, "
                 f"const T* __restrict__ {b_name}, const T* __restrict__ w, T* __restrict__ {out_name}) {{ "
                 f"{name}4<T, T, T, T>({a_name}, {b_name}, w, {out_name}); }}")
    return "\n".join(lines)


def emit_bx12(nz, d1, d2):
    """Fused reverse of both operands: g1[i] = sum w[p] c go[k] x2[j] and g2[j] = sum w[p] c go[k] x1[i] share the
    factor t = c * (w[p] * go[k]) of every non-zero: 3 multiply-adds per non-zero instead of 4 (+ one product per
    (k, p) pair).  go / x1 / g1 / g2 may be 2-vectors (packed two-edge evaluation), x2 and w are per-lane scalars."""
    lines = ["  template <typename TG, typename TX2, typename TW> static __device__ __forceinline__ void "
             "bx12(const TG* __restrict__ go, const TG* __restrict__ x1, const TX2* __restrict__ x2, "
             "const TW* __restrict__ w, TG* __restrict__ g1, TG* __restrict__ g2) {"]
    seen1, seen2 = set(), set()
    groups = {}
    for (i, j, k, p, v) in nz:
        groups.setdefault((k, p), []).append((i, j, v))
    for (k, p) in sorted(groups):
        lines.append(f"    {{ const TG gw = go[{k}] * w[{p}];")
        for (i, j, v) in sorted(groups[(k, p)]):
            lines.append(f"      {{ const TG t = gw * TW({v!r}); g1[{i}] {'+=' if i in seen1 else '='} t * x2[{j}]; "
                         f"g2[{j}] {'+=' if j in seen2 else '='} t * x1[{i}]; }}")
            seen1.add(i)
            seen2.add(j)
        lines.append("    }")
    for i in range(d1):
        if i not in seen1:
            lines.append(f"    g1[{i}] = TG(0);")
    for j in range(d2):
        if j not in seen2:
            lines.append(f"    g2[{j}] = TG(0);")
    lines.append("  }")
    return "\n".join(lines)


def main(out_path=None):
    out_path = out_path or os.path.join(HERE, "aa_cg_gen.h")
    sigs = signatures()
    o = ["// GENERATED by allegro_amd/csrc/gen_cg.py -- do not edit.",
         "// Straight-line Clebsch-Gordan contractions (real basis, component normalisation folded) for the",
         "// tensor-product signatures of standard Allegro stacks; see gen_cg.py.",
         "#pragma once", "#include <hip/hip_runtime.h>", "", "namespace aa {", "namespace cg {", ""]
    for n, s in enumerate(sigs):
        d1, d2, dout = s["dims"]
        nz = s["nz"]
        o.append(f"// {s['name']}  ({s['P']} paths, {len(nz)} non-zeros{', ij-diagonal' if s['diag'] else ''})")
        o.append(f"struct Sig{n} {{")
        o.append(f"  static constexpr int LMAX = {s['lmax']}, D1 = {d1}, D2 = {d2}, DOUT = {dout}, P = {s['P']}, NNZ = {len(nz)};")
        o.append(emit_fn("fwd", "x1", "x2", "out", dout, [(k, p, i, j, v) for (i, j, k, p, v) in nz]))
        o.append(emit_fn("bx1", "go", "x2", "g1", d1, [(i, p, k, j, v) for (i, j, k, p, v) in nz]))
        o.append(emit_fn("bx2", "go", "x1", "g2", d2, [(j, p, k, i, v) for (i, j, k, p, v) in nz]))
        o.append(emit_bx12(nz, d1, d2))
        o.append("};")
        o.append(f"static const int sig{n}_nz[][4] = {{" + ", ".join(f"{{{i},{j},{k},{p}}}" for (i, j, k, p, v) in nz) + "};")
        o.append(f"static const double sig{n}_val[] = {{" + ", ".join(repr(v) for (i, j, k, p, v) in nz) + "};")
        o.append("")
    o.append(f"constexpr int kNumSigs = {len(sigs)};")
    o.append("struct SigInfo { int lmax, d1, d2, dout, num_paths, nnz; const int (*nz)[4]; const double* val; };")
    o.append("static const SigInfo kSigs[kNumSigs] = {")
    for n, s in enumerate(sigs):
        d1, d2, dout = s["dims"]
        o.append(f"  {{{s['lmax']}, {d1}, {d2}, {dout}, {s['P']}, {len(s['nz'])}, sig{n}_nz, sig{n}_val}},")
    o.append("};")
    o.append("#define AA_FOREACH_SIG(X) " + " ".join(f"X({n}, Sig{n})" for n in range(len(sigs))))
    o += ["", "}  // namespace cg", "}  // namespace aa", ""]
    with open(out_path, "w") as f:
        f.write("\n".join(o))
    return out_path


if __name__ == "__main__":
    print(main())
