#!/usr/bin/env python3
"""Generates icicle_amd/csrc/mont_asm.hpp: the interleaved Montgomery product of bigfield.hpp (radix 2^29, product
scanning) written as ONE gfx950 inline-asm block per operation, for N = 9 (BN254) and N = 14 (BLS12-381) limbs.

Why: hipcc reassociates every column sum so that the carry of the previous column is added LAST (LLVM's Reassociate
ranks it highest), which costs one v_lshl_add_u64 per column on top of the v_mad_u64_u32 chain -- 16 of the 220
instructions of a BN254 product (profiles/r01_notes.md tried single-instruction asm barriers: the hazard recogniser
answers each with an s_nop). Here the carry is the addend of the column's first v_mad_u64_u32, so a product is
2*N*N mads + (2N-2) shifts + N (mul_lo + and) + N masks/alignbit = 205 instructions for N = 9.

The 64-bit column accumulator lives in the fixed pair v[2:3] (inline asm cannot name the halves of a 64-bit operand;
the register allocator simply keeps v2/v3 free across the block). Carry-out of v_mad_u64_u32 goes to vcc (never read).
Operands: %0..%(N-1) = r (early-clobber outputs), then m[N] scratch outputs, then the inputs.
"""
import sys

ACC = "v[2:3]"
LO, HI = "v2", "v3"
RB = 29
MASK = "0x1fffffff"


def gen(n: int, kind: str) -> str:
    """kind: mul (a*b), mul_add (a*b + c*d), sqr (a*a; inputs a and a2 = 2a)"""
    ops = []  # textual operand list in order
    r = [f"%{i}" for i in range(n)]
    m = [f"%{n + i}" for i in range(n)]
    base = 2 * n + 1  # + acc
    nin = {"mul": 2, "mul_add": 4, "sqr": 2}[kind]
    vin = [[f"%{base + j * n + i}" for i in range(n)] for j in range(nin)]
    p = [f"%{base + nin * n + i}" for i in range(n)]
    pinv = f"%{base + nin * n + n}"
    lines = []
    first = [True]

    def mad(x, y):
        lines.append(f"v_mad_u64_u32 {ACC}, vcc, {x}, {y}, {'0' if first[0] else ACC}")
        first[0] = False

    def products(k):
        lo = max(0, k - n + 1)
        hi = min(k, n - 1)
        if kind == "sqr":
            a, a2 = vin
            for i in range(lo, hi + 1):
                j = k - i
                if i < j:
                    mad(a2[i], a[j])
                elif i == j:
                    mad(a[i], a[i])
        else:
            for pair in range(nin // 2):
                x, y = vin[2 * pair], vin[2 * pair + 1]
                for i in range(lo, hi + 1):
                    mad(x[i], y[k - i])

    for k in range(2 * n - 1):
        products(k)
        if k < n:
            for i in range(k):
                mad(m[i], p[k - i])
            lines.append(f"v_mul_lo_u32 {m[k]}, {LO}, {pinv}")
            lines.append(f"v_and_b32 {m[k]}, {MASK}, {m[k]}")
            mad(m[k], p[0])
            lines.append(f"v_lshrrev_b64 {ACC}, {RB}, {ACC}")
        else:
            for i in range(k - n + 1, n):
                mad(m[i], p[k - i])
            lines.append(f"v_and_b32 {r[k - n]}, {MASK}, {LO}")
            if k < 2 * n - 2:
                lines.append(f"v_lshrrev_b64 {ACC}, {RB}, {ACC}")
            else:
                lines.append(f"v_alignbit_b32 {r[n - 1]}, {HI}, {LO}, {RB}")
    return lines


def emit_inplace(n: int) -> str:
    """a <- a*b with the result in a's own registers: a[k-N] is dead before r[k-N] is written (column k only reads
    a[i], i > k-N), so r and a can be the same read-write operands -- an accumulator coordinate updated in a loop then
    needs no register-to-register copies at the back edge (early-clobber outputs can never be coalesced with inputs)."""
    lines = gen(n, "mul")
    # operand renumbering: gen() numbers r = %0..%(n-1), m = %n.., acc, a = %(2n+1).., b, p, pinv; here a IS r
    base = 2 * n + 1
    ren = {}
    for i in range(n):
        ren[f"%{base + i}"] = f"%{i}"          # a[i] -> r[i]
    shift = n
    for j in range(base + n, base + 3 * n + 1):  # b, p, pinv move down by n
        ren[f"%{j}"] = f"%{j - shift}"
    import re
    out = []
    for ln in lines:
        out.append(re.sub(r"%\d+", lambda mo: ren.get(mo.group(0), mo.group(0)), ln))
    body = "\n".join(f'      "{ln}\\n"' for ln in out)
    outs = ", ".join([f'"+&v"(a[{i}])' for i in range(n)] + [f'"=&v"(m[{i}])' for i in range(n)] + ['"=&{v[2:3]}"(acc)'])
    ins = ", ".join([f'"v"(b[{i}])' for i in range(n)] + [f'"s"(PR::P[{i}])' for i in range(n)] + ['"s"(PR::PINV)'])
    return f"""  // mul in place (a <- a*b), N = {n}
  template <class PR>
  __device__ __forceinline__ void mont_mul_inplace_asm{n}(uint32_t* a, const uint32_t* b)
  {{
    uint32_t m[{n}];
    uint64_t acc;
    asm(
{body}
      : {outs}
      : {ins}
      : "vcc");
    (void)acc;
  }}
"""


def emit_mul_add_inplace_c(n: int) -> str:
    """c <- a*b + c*d with the result in c's own registers (same argument as emit_inplace: c[k-N] is dead before
    r[k-N] is written)."""
    import re
    lines = gen(n, "mul_add")
    base = 2 * n + 1  # gen(): r %0.., m %n.., acc, a, b, c, d, p, pinv
    ren = {}
    for i in range(n):
        ren[f"%{base + 2 * n + i}"] = f"%{i}"  # c[i] -> r[i]
    for j in range(base + 3 * n, base + 5 * n + 1):  # d, p, pinv move down by n
        ren[f"%{j}"] = f"%{j - n}"
    out = [re.sub(r"%\d+", lambda mo: ren.get(mo.group(0), mo.group(0)), ln) for ln in lines]
    body = "\n".join(f'      "{ln}\\n"' for ln in out)
    outs = ", ".join([f'"+&v"(c[{i}])' for i in range(n)] + [f'"=&v"(m[{i}])' for i in range(n)] + ['"=&{v[2:3]}"(acc)'])
    ins = ", ".join([f'"v"({nm}[{i}])' for nm in ("a", "b", "d") for i in range(n)] + [f'"s"(PR::P[{i}])' for i in range(n)] + ['"s"(PR::PINV)'])
    return f"""  // mul_add in place (c <- a*b + c*d), N = {n}
  template <class PR>
  __device__ __forceinline__ void mont_mul_add_inplace_c_asm{n}(uint32_t* c, const uint32_t* a, const uint32_t* b, const uint32_t* d)
  {{
    uint32_t m[{n}];
    uint64_t acc;
    asm(
{body}
      : {outs}
      : {ins}
      : "vcc");
    (void)acc;
  }}
"""


def emit(n: int, kind: str) -> str:
    lines = gen(n, kind)
    body = "\n".join(f'      "{ln}\\n"' for ln in lines)
    nin = {"mul": 2, "mul_add": 4, "sqr": 2}[kind]
    names = {"mul": ["a", "b"], "mul_add": ["a", "b", "c", "d"], "sqr": ["a", "a2"]}[kind]
    outs = ", ".join([f'"=&v"(r[{i}])' for i in range(n)] + [f'"=&v"(m[{i}])' for i in range(n)] + ['"=&{v[2:3]}"(acc)'])
    ins = ", ".join([f'"v"({nm}[{i}])' for nm in names for i in range(n)] + [f'"s"(PR::P[{i}])' for i in range(n)] + ['"s"(PR::PINV)'])
    args = ", ".join(f"const uint32_t* {nm}" for nm in names)
    nmads = sum(1 for ln in lines if ln.startswith("v_mad"))
    return f"""  // {kind}, N = {n}: {len(lines)} instructions ({nmads} v_mad_u64_u32)
  template <class PR>
  __device__ __forceinline__ void mont_{kind}_asm{n}(uint32_t* r, {args})
  {{
    uint32_t m[{n}];
    uint64_t acc;
    asm(
{body}
      : {outs}
      : {ins}
      : "vcc");
    (void)acc;
  }}
"""


def main():
    out = ["// GENERATED by tools/gen_mont_asm.py -- do not edit. See that file for the why and the operand layout.",
           "#pragma once", "#include <cstdint>", "#if defined(__HIPCC__)", "  #include <hip/hip_runtime.h>", "#endif", "#if defined(__HIP_DEVICE_COMPILE__)", "namespace icicle_hip {", ""]
    for n in (9, 14):
        for kind in ("mul", "sqr", "mul_add"):
            out.append(emit(n, kind))
        out.append(emit_inplace(n))
        out.append(emit_mul_add_inplace_c(n))
    out += ["} // namespace icicle_hip", "#endif", ""]
    sys.stdout.write("\n".join(out))


if __name__ == "__main__":
    main()
