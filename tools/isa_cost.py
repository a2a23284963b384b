"""Static cost of a kernel's basic blocks from `hipcc -S` output (make -C raytracer_amd/csrc isa ISADIR=...).

    python tools/isa_cost.py file.s kernel-name-substring [first-line last-line]

Prints one line per basic block (label, source line of the .s file, instruction counts by class and the vector-ALU
issue cycles they stand for at the rates of profiles/valu_issue_rates.txt: 2.8 cycles for the plain two-operand and
FMA forms on vector registers, 4.7 for compares, selects, maxima, conversions and three-operand integer forms, 8.2 for
the transcendental unit, 4 per matrix instruction), then the totals.  With a line range, only that part of the file."""
import re
import sys

FAST = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32",
        "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_not_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_f64", "v_rsq_f64",
         "v_rcp_iflag_f32"}


def classify(op, operands):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if base in TRANS:
            return "trans"
        if base in FAST and not re.search(r"(^|[ ,])(s\[?\d|s\d|vcc|exec)", operands):
            return "fast"
        return "slow"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


COST = {"fast": 2.8, "slow": 4.7, "trans": 8.2, "mfma": 4.0}


def main():
    path, name = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and name in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []
    cur = {"label": lines[start][:-1][-24:], "line": start + 1, "n": {}, "marks": []}
    for i in range(start + 1, end):
        l = lines[i].strip()
        if not l or l.startswith(("//", ";")) and "MARK" not in l:
            continue
        l = l.split(" ; ")[0].strip() if not l.startswith(";") else l
        if l.startswith(";") and "MARK" in l:
            cur["marks"].append(l.split("MARK", 1)[1].strip())
            continue
        if l.startswith("."):
            if l.endswith(":") or re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur)
                cur = {"label": l.split(":")[0], "line": i + 1, "n": {}, "marks": []}
            continue
        parts = l.split(None, 1)
        op = parts[0]
        c = classify(op, parts[1] if len(parts) > 1 else "")
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if op.startswith("s_memtime"):
            cur["marks"].append("memtime")
    blocks.append(cur)
    tot = {}
    print("%-12s %6s %5s %5s %5s %5s %4s %4s %4s %4s %7s  marks" % ("block", "line", "fast", "slow", "trans", "mfma", "lds", "vmem", "salu", "br", "cycles"))
    for b in blocks:
        if not (lo <= b["line"] <= hi):
            continue
        n = b["n"]
        cyc = sum(COST[k] * n.get(k, 0) for k in COST)
        for k, v in n.items():
            tot[k] = tot.get(k, 0) + v
        print("%-12s %6d %5d %5d %5d %5d %4d %4d %4d %4d %7.0f  %s" % (b["label"], b["line"], n.get("fast", 0), n.get("slow", 0), n.get("trans", 0),
              n.get("mfma", 0), n.get("lds", 0), n.get("vmem", 0), n.get("salu", 0), n.get("branch", 0), cyc, " ".join(b["marks"])))
    print("total", tot, "valu cycles %.0f" % sum(COST[k] * tot.get(k, 0) for k in COST))


if __name__ == "__main__":
    main()
