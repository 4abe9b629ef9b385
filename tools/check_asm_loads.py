#!/usr/bin/env python3
"""Disassembly-level guard for the hand-pipelined loads (chain_kernel.hpp; until round 4 also the channel-pair sweep, retired).

Those kernels issue `global_load_*` from inline asm a trip or a row group ahead and wait with a hand-written
`s_waitcnt vmcnt(0)`.  An asm statement's output operand tells hipcc the value EXISTS when the statement ends, so under
register pressure it may copy the "value" elsewhere (a v_mov, a v_accvgpr_write) or reuse the register while the load is
still in flight: wrong pixels, then a load landing on a register that by now holds an address -- the fault of round 2.
The property that rules this out is checked here on the machine code itself, for EVERY vector-memory load
of every kernel in the given code objects (the compiler's own loads satisfy it by construction, so checking all of them
costs nothing and needs no way to tell the two kinds apart):

    between a load into v[a:b] and the s_waitcnt that retires it, no instruction reads or writes v[a:b]
    (nor an AGPR copy of it), on any path through the kernel.

Model: vmcnt counts vector-memory instructions in issue order (loads and, on gfx9 targets, stores); `s_waitcnt vmcnt(N)`
retires all but the N youngest.  Paths: a forward may-analysis over the control-flow graph (every branch both ways, loops
to their fixed point), path-insensitive: a combination of branches that the scalar conditions never take together still
counts -- the rule must hold for the code as laid out, not for the values it happens to run on.  Register-indexed moves (s_set_gpr_idx_on) touch an unknown
register of their array: they are taken to touch the IDX_SPAN registers from their base.
(A wave that ends with a load in flight is not an error here: hipcc's own early exits do that with loads whose results
are dead, and the hardware holds the wave's registers until its memory instructions have returned.)

usage: check_asm_loads.py [--verbose] <object.hip.o | code object> ...     exit 1 when a kernel breaks the rule
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
IDX_SPAN = 32           # registers a GPR-indexed operand may reach from its base (the longest register vector in the kernels)
# Kernels that issue loads from inline asm.  A jump table (s_setpc_b64, hipcc's code for a dense switch) cannot be
# followed; in a kernel of this list that is an error, elsewhere the blocks behind it are simply not walked (every load
# there is the compiler's own, waited for by the compiler).
HAND_PIPELINED = re.compile(r"k_chainILi")
# kernels whose indexed register vector is shorter than IDX_SPAN: (name pattern, span from the match); none at present
IDX_SPAN_OF = []
VMCNT_MAX = 63

_reg_re = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_func_re = re.compile(r"^([0-9a-f]+) <(\S+)>:\s*$")
_ins_re = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):\s+[0-9A-Fa-f ]+?(?:<(\S+?)\+0x([0-9a-f]+)>|<(\S+?)>)?\s*$")
_VMEM = re.compile(r"^(global|buffer|flat|scratch|tbuffer|image)_")


def regs_of(text):
    out = set()
    for m in _reg_re.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            out.update((kind, r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "mnem", "ops", "target", "line")

    def __init__(self, addr, mnem, ops, target, line):
        self.addr, self.mnem, self.ops, self.target, self.line = addr, mnem, ops, target, line


def parse(disasm):
    """-> {function name: [Ins]}"""
    funcs, cur, start = {}, None, 0
    for line in disasm.splitlines():
        m = _func_re.match(line)
        if m:
            start = int(m.group(1), 16)
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None:
            continue
        m = _ins_re.match(line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2) or "", int(m.group(3), 16)
        target = None
        if mnem.startswith(("s_cbranch", "s_branch")):
            if m.group(5) is not None:
                target = start + int(m.group(5), 16)
            elif m.group(6) is not None:
                target = start
        cur.append(Ins(addr, mnem, ops, target, line.strip()))
    return funcs


def split_operands(ops):
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _analyse(name, body, use_flags):
    """-> list of violation strings.
    Forward may-analysis over the kernel's control-flow graph.  State at an instruction: for every register with a load
    possibly in flight into it, the SMALLEST number of vector-memory instructions issued after that load on any path
    (and where the load was issued).  `s_waitcnt vmcnt(N)` retires a load once N or more were issued after it, so the
    smallest count is the safe one; joining paths takes the union of the registers and the minimum of the counts."""
    n = len(body)
    index = {ins.addr: i for i, ins in enumerate(body)}
    succs, touched, kind = [None] * n, [None] * n, [None] * n
    for i, ins in enumerate(body):
        mnem, ops = ins.mnem, ins.ops
        if mnem == "s_endpgm":
            succs[i] = ()
        elif mnem in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            succs[i] = "indirect"                       # a jump table (hipcc's own, from a switch): not followed, see below
        elif mnem == "s_branch" and ins.target is not None:
            succs[i] = (index[ins.target],) if ins.target in index else ()
        elif mnem.startswith("s_cbranch") and ins.target is not None:
            succs[i] = tuple(x for x in ((index.get(ins.target)), i + 1) if x is not None and x < n)
        else:
            succs[i] = (i + 1,) if i + 1 < n else ()
        touched[i] = frozenset(regs_of(ops))
        if _VMEM.match(mnem) and "_load" in mnem:
            # a load INTO a register with an older load in flight is in order (loads return in issue order): only the
            # address operands of a load count as touches
            touched[i] = frozenset(regs_of(",".join(split_operands(ops)[1:])))
        if mnem == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            kind[i] = ("wait", int(m.group(1))) if m else None
        elif mnem == "s_set_gpr_idx_on":
            m = re.search(r"gpr_idx\(([^)]*)\)", ops)
            bits = 0
            for part in (m.group(1).split(",") if m else []):
                bits |= {"SRC0": 1, "SRC1": 2, "SRC2": 4, "DST": 8}.get(part.strip(), 0)
            kind[i] = ("idx", bits or 15)
        elif mnem == "s_set_gpr_idx_off":
            kind[i] = ("idx", 0)
        elif mnem == "s_set_gpr_idx_mode":
            kind[i] = ("idx", 15)
        elif _VMEM.match(mnem):
            operands = split_operands(ops)
            is_load = "_load" in mnem or ("_atomic" in mnem and re.search(r"\b(glc|sc0)\b", ops))
            dst = frozenset()
            if is_load and not re.search(r"\blds\b", ops) and operands:
                dst = frozenset(regs_of(operands[0]))
            kind[i] = ("vmem", dst)
        elif mnem == "s_endpgm":
            kind[i] = ("end",)

    indirect = any(x == "indirect" for x in succs)
    succs = [() if x == "indirect" else x for x in succs]
    span = IDX_SPAN
    for pat, fn in IDX_SPAN_OF:
        m_ = pat.search(name)
        if m_:
            span = min(IDX_SPAN, fn(m_))

    def indexed_touch(i, idx_mode):
        out = set()
        for pos, op in enumerate(split_operands(body[i].ops)):
            if idx_mode & (8 if pos == 0 else (1 << (pos - 1))):
                for k_, r in regs_of(op):
                    if k_ == "v":
                        out.update(("v", r + k) for k in range(span))
        return out

    # Scalar flags.  hipcc's structurizer routes a loop's exits through shared blocks: "s_mov_b64 s[a:b], -1" on the way
    # out, "s_and_b64 vcc, exec, s[a:b]; s_cbranch_vccnz <exit>" in the shared block.  Followed blindly, such a block
    # lets the exit path run on into the loop body.  So the walk keeps what it KNOWS about 64-bit scalar flags (only
    # constants 0 / -1 written by s_mov_b64, and vcc derived from them with exec taken as non-zero -- these are
    # wave-uniform loops) and keeps program points apart by that knowledge; a conditional branch on a known vcc is
    # followed one way only.  Anything else that names a scalar register forgets it.
    sreg_re = re.compile(r"\b(?:s(\d+)|s\[(\d+):(\d+)\]|(vcc))\b")

    def sregs_of(text):
        out = set()
        for m in sreg_re.finditer(text):
            if m.group(4):
                out.add("vcc")
            elif m.group(1) is not None:
                out.add(int(m.group(1)))
            else:
                out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    flag_op = [None] * n
    tested = set()
    for ins in body:
        ops = split_operands(ins.ops)
        if ins.mnem in ("s_and_b64", "s_andn2_b64") and len(ops) == 3 and ops[0] == "vcc" and ops[1] == "exec":
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", ops[2])
            if m:
                tested.add(int(m.group(1)))
    for i, ins in enumerate(body):
        if not use_flags:
            break
        ops = split_operands(ins.ops)
        if ins.mnem == "s_mov_b64" and len(ops) == 2 and ops[1] in ("-1", "0"):
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", ops[0])
            if m and int(m.group(1)) in tested:
                flag_op[i] = ("set", int(m.group(1)), -1 if ops[1] == "-1" else 0)
            elif ops[0] == "vcc":
                flag_op[i] = ("setvcc", "nz" if ops[1] == "-1" else "z")
        elif ins.mnem in ("s_and_b64", "s_andn2_b64") and len(ops) == 3 and ops[0] == "vcc" and ops[1] == "exec":
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", ops[2])
            if m:
                flag_op[i] = ("test", int(m.group(1)), ins.mnem == "s_andn2_b64")
        elif ins.mnem in ("s_cbranch_vccnz", "s_cbranch_vccz"):
            flag_op[i] = ("br", ins.mnem == "s_cbranch_vccnz")
        if flag_op[i] is None:
            named = sregs_of(ins.ops)
            if ins.mnem.startswith(("v_cmp", "v_div_scale")) or "_co_" in ins.mnem:
                named.add("vcc")
            if ins.mnem.startswith("s_cbranch") or ins.mnem in ("s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm"):
                named = set()
            flag_op[i] = ("kill", frozenset(named)) if named else None

    # state per (instruction, known flags): (dict reg -> (count, frozenset(origin addresses)), idx_mode bits)
    IN = [dict() for _ in range(n)]
    IN[0][frozenset()] = ({}, 0)
    work, queued = [(0, frozenset())], {(0, frozenset())}
    problems, reported = [], set()
    visits = 0
    while work:
        i, facts = work.pop()
        queued.discard((i, facts))
        regs, idx_mode = IN[i][facts]
        visits += 1
        if visits > 2_000_000:
            problems.append("%s: too many states" % name)
            break
        ins = body[i]
        t = touched[i]
        if idx_mode and ins.mnem.startswith("v_"):
            t = t | indexed_touch(i, idx_mode)
        k = kind[i]
        if regs:
            hit = [r for r in t if r in regs]
            if hit and (i, tuple(sorted(hit))) not in reported:
                reported.add((i, tuple(sorted(hit))))
                origins = sorted({a for r in hit for a in regs[r][1]})
                problems.append("%s: 0x%x  %s  touches %s while the load issued at %s may still be in flight" % (
                    name, ins.addr, ins.line.split("//")[0].strip(), ", ".join("%s%d" % r for r in sorted(hit)),
                    ", ".join("0x%x" % a for a in origins)))
        out = regs
        if k is not None:
            if k[0] == "wait":
                out = {r: v for r, v in regs.items() if v[0] < k[1]}
            elif k[0] == "idx":
                idx_mode = k[1]
            elif k[0] == "vmem":
                out = {r: (min(v[0] + 1, VMCNT_MAX + 1), v[1]) for r, v in regs.items()}
                for r in k[1]:
                    out[r] = (0, frozenset([ins.addr]))
        nexts = succs[i]
        f = flag_op[i]
        if f is not None:
            d = dict(facts)
            if f[0] == "set":
                d[f[1]] = f[2]; d.pop(f[1] + 1, None)
            elif f[0] == "setvcc":
                d["vcc"] = f[1]
            elif f[0] == "test":
                v = d.get(f[1])
                d.pop("vcc", None)
                if v is not None:
                    d["vcc"] = ("z" if v == -1 else "nz") if f[2] else ("nz" if v == -1 else "z")
            elif f[0] == "kill":
                for r in f[1]:
                    d.pop(r, None)
                    if isinstance(r, int):
                        d.pop(r - 1, None)          # the pair that starts one below covers it too
            elif f[0] == "br":
                v = d.get("vcc")
                if v is not None and len(nexts) == 2:
                    taken = (v == "nz") == f[1]
                    nexts = (nexts[0],) if taken else (nexts[1],)
            facts_out = frozenset(d.items())
        else:
            facts_out = facts
        for s_ in nexts:
            cur = IN[s_].get(facts_out)
            if cur is None:
                IN[s_][facts_out] = (dict(out), idx_mode)
                changed = True
            else:
                cregs, cidx = cur
                changed = False
                for r, v in out.items():
                    c = cregs.get(r)
                    if c is None:
                        cregs[r] = v; changed = True
                    elif v[0] < c[0] or not v[1] <= c[1]:
                        cregs[r] = (min(v[0], c[0]), c[1] | v[1]); changed = True
                if idx_mode | cidx != cidx:
                    IN[s_][facts_out] = (cregs, cidx | idx_mode); changed = True
            if changed and (s_, facts_out) not in queued:
                queued.add((s_, facts_out))
                work.append((s_, facts_out))
    if indirect and HAND_PIPELINED.search(name) and not any("indirect" in p_ for p_ in problems):
        problems.append("%s: indirect control flow (s_setpc_b64) in a kernel with hand-written loads: cannot be verified" % name)
    return problems, visits


def check_function(name, body, verbose=False):
    """-> list of violation strings.  First without any knowledge of scalar flags (every branch both ways: cheap, and
    enough for kernels whose loads hipcc placed itself); a kernel that fails that way is walked again with the flags."""
    problems, visits = _analyse(name, body, False)
    if problems:
        problems, visits = _analyse(name, body, True)
    if verbose:
        loads = sum(1 for ins in body if _VMEM.match(ins.mnem) and "_load" in ins.mnem)
        print("  %-92s %5d instructions, %3d loads, %6d visits: %s" % (name[:92], len(body), loads, visits, "ok" if not problems else "BROKEN"))
    return problems


def code_objects(path, tmp):
    """A host object with an offload bundle -> its gfx950 code objects; anything else is taken as a code object."""
    if path.endswith(".o"):
        local = os.path.join(tmp, os.path.basename(path))
        with open(path, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "-d", "--offloading", os.path.basename(local)], cwd=tmp,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return sorted(glob.glob(local + "*gfx950"))
    return [path]


def check_paths(paths, verbose=False, only=None):
    """-> (number of kernels checked, [violations])"""
    checked, problems = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for path in paths:
            for co in code_objects(path, tmp):
                text = subprocess.run([OBJDUMP, "-d", co], stdout=subprocess.PIPE, text=True).stdout
                for name, body in parse(text).items():
                    if not body or (only and not re.search(only, name)):
                        continue
                    checked += 1
                    problems += check_function(name, body, verbose)
    return checked, problems


def main(argv):
    verbose = "--verbose" in argv
    paths = [a for a in argv[1:] if not a.startswith("--")]
    if not paths:
        print(__doc__)
        return 2
    if not os.path.exists(OBJDUMP):
        print("check_asm_loads: %s not found" % OBJDUMP)
        return 2
    checked, problems = check_paths(paths, verbose)
    for p in problems:
        print("check_asm_loads: " + p)
    print("check_asm_loads: %d kernels, %d violations" % (checked, len(problems)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
