"""Per-basic-block instruction histogram of one kernel's ISA (hipcc -S --cuda-device-only): which loop bodies hold the
vector / matrix / LDS instructions.   python tools/isa_blocks.py kernel.s [min_instructions]"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    floor = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    blocks, cur = [], None
    for l in lines:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            cur = [m.group(1), collections.Counter(), 0]
            blocks.append(cur)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if cur is None:
            cur = ["entry", collections.Counter(), 0]
            blocks.append(cur)
        cur[1][op] += 1
        cur[2] += 1
        if op.startswith("s_cbranch") or op == "s_branch":          # the fall-through code is a block of its own
            cur = [cur[0] + "+", collections.Counter(), 0]
            blocks.append(cur)
    print(len(blocks), "blocks,", sum(b[2] for b in blocks), "instructions")
    print("block            total  mfma  valu   exp   mix cvtpk    ds  glob  snop swait  sbar  perm  vmov")
    for name, c, n in blocks:
        if n < floor:
            continue
        g = lambda pred: sum(v for k, v in c.items() if pred(k))
        print(f"{name:15s} {n:6d} {g(lambda k: 'mfma' in k):5d} {g(lambda k: k.startswith('v_') and 'mfma' not in k):5d} "
              f"{c.get('v_exp_f32', 0):5d} {g(lambda k: 'fma_mix' in k):5d} {c.get('v_cvt_pk_f16_f32', 0):5d} "
              f"{g(lambda k: k.startswith('ds_')):5d} {g(lambda k: k.startswith('global_')):5d} {c.get('s_nop', 0):5d} "
              f"{c.get('s_waitcnt', 0):5d} {c.get('s_barrier', 0):5d} {g(lambda k: 'permlane' in k):5d} "
              f"{g(lambda k: k.startswith('v_mov') or k.startswith('v_accvgpr')):5d}")
    if len(sys.argv) > 3:
        for name, c, n in blocks:
            if name == sys.argv[3]:
                for k, v in c.most_common(40):
                    print(f"   {k:28s} {v}")


if __name__ == "__main__":
    main()
