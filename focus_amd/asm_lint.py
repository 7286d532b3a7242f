"""Lint of kernels that issue global loads from inline asm with hand-counted s_waitcnt (cdna_hip_programming.md 5.7 form
(iii); traj_time2.hip, slot_attn.hip).  The compiler does not know such a load is still in flight: if it copies, spills,
parks in an AGPR or otherwise reads the destination before the counted wait, it reads whatever the register held before --
only when the load is late, so a green GPU run does not show it.  Seen three times: an AGPR park in
slot_bwd_defer_kernel<8, false, true>, an address register overwritten by a late load, and v_mov rotations of ring
registers on a loop back edge in time2_dx_lds_kernel (after an unrelated change of the loop's exit).

lint_hand_loads(device .s) walks every kernel that has such loads along all feasible paths of its control-flow graph with
  * the queue of vector-memory loads not yet retired by an `s_waitcnt vmcnt(N)` (loads retire in order; stores are left out
    because they may retire out of order with loads and so never make a wait stricter; compiler-issued and LDS-DMA loads
    take a queue position but carry no registers of interest),
  * the set of registers holding values DERIVED from a register that was read while in flight ("poison"),
and reports: scratch use; any write to a register under an in-flight hand load; any in-flight or poisoned register reaching
an effect (store, LDS write, address of a load, scalar/vcc result).  A poisoned value that dies unused is not reported.
Path feasibility is tracked just far enough for hipcc's structured control flow: exit flags (`s_mov_b64 s[a:b], -1` ...
`s_and_b64 vcc, exec, s[a:b]` ... `s_cbranch_vccnz`) and exec known non-empty (`s_cbranch_execnz` after exec was restored).
States are memoised per (block, queue, poison, flags): loops are followed until the state repeats."""
import re
import sys

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_VM_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)_")
_EFFECT = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|ds_write|ds_store|global_atomic|buffer_atomic|flat_atomic|"
                     r"ds_add|ds_max|ds_min|ds_bpermute|ds_permute|v_cmp|v_cmpx|v_readlane|v_readfirstlane)")
_KEEPS = re.compile(r"fmac|_mac_|UNUSED_PRESERVE|op_sel|v_dot\w*c_|v_cvt_scalef32_pk_fp8|v_cvt_pk_fp8|v_cvt_sr_|sdwa|_dpp|v_writelane")
_SREG = re.compile(r"^s(\d+)$|^s\[(\d+):(\d+)\]$")


def _regs(text):
    """VGPRs (n) and AGPRs (1000 + n) named in an operand string."""
    regs = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(2)) + (1000 if m.group(1) == "a" else 0))
        else:
            base = 1000 if m.group(3) == "a" else 0
            regs.update(range(base + int(m.group(4)), base + int(m.group(5)) + 1))
    return regs


def _srange(name):
    m = _SREG.match(name)
    if not m:
        return None
    return (int(m.group(1)),) * 2 if m.group(1) is not None else (int(m.group(2)), int(m.group(3)))


def _kernels(asm_path):
    """[(name, [(text, in_asm_block)], scratch bytes)] for every kernel of a device .s file."""
    out, name, body, in_asm, pending = [], None, [], False, None
    with open(asm_path) as f:
        for line in f:
            s = line.strip()
            if name is None:
                if pending is not None and s.startswith("; ScratchSize:"):
                    out.append((pending[0], pending[1], int(s.split(":")[1])))
                    pending = None
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    name, body, in_asm = m.group(1), [], False
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif s.startswith(".Lfunc_end"):
                pending, name = (name, body), None
            else:
                t = s.split(";", 1)[0].strip()
                if t:
                    body.append((t, in_asm))
    return out


def lint_kernel(body):
    labels = {t[:-1]: i for i, (t, _) in enumerate(body) if t.endswith(":")}
    problems, seen = set(), set()
    work = [(0, (), frozenset(), (), None, True)]
    while work:
        i, queue, poison, consts, vcc, live = work.pop()
        queue, poison, consts = list(queue), set(poison), dict(consts)
        while i < len(body):
            t, in_asm = body[i]
            if t.endswith(":"):
                while queue and not queue[0]:
                    queue.pop(0)
                key = (i, tuple(queue), frozenset(poison), tuple(sorted(consts.items())), vcc, live)
                if key in seen:
                    break
                seen.add(key)
                i += 1
                continue
            if t.startswith("."):
                i += 1
                continue
            parts = t.split(None, 1)
            op, args = parts[0], parts[1] if len(parts) > 1 else ""
            ops = [a.strip() for a in args.split(",")]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    del queue[:max(0, len(queue) - int(m.group(1)))]
                i += 1
                continue
            if op == "s_branch" or op.startswith("s_cbranch"):
                tgt = args.strip()
                if tgt in labels:
                    taken = {"s_cbranch_vccnz": vcc, "s_cbranch_vccz": None if vcc is None else 1 - vcc,
                             "s_cbranch_execnz": 1 if live else None, "s_cbranch_execz": 0 if live else None}.get(op)
                    if op == "s_branch" or taken == 1:
                        i = labels[tgt]
                        continue
                    if taken is None:
                        work.append((labels[tgt], tuple(queue), frozenset(poison), tuple(consts.items()), vcc, live))
                i += 1
                continue
            if op == "s_endpgm":
                break
            flying = frozenset().union(*queue) if queue else frozenset()
            dirty = flying | poison
            effect = bool(_EFFECT.match(op))
            dst = set() if effect and not op.startswith(("v_cmp", "v_read", "ds_bpermute", "ds_permute")) else _regs(ops[0] if ops else "")
            src = _regs(args) if not dst else _regs(",".join(ops[1:]))
            if op.startswith("v_swap"):
                src |= dst
            if dst & flying:
                problems.add("%s writes %s under a hand-issued load still in flight" % (op, _names(dst & flying)))
            hit = src & dirty
            sdst = bool(ops) and (ops[0].startswith(("s", "vcc", "exec", "m0")) and not ops[0].startswith("src"))
            if hit and (effect or sdst or _VM_LOAD.match(op)):
                problems.add("%s uses %s: %s" % (op, _names(hit), "loaded by hand and still in flight" if hit & flying else
                                                 "derived from a register read while its hand-issued load was in flight"))
            elif hit or (dst & poison and _KEEPS.search(t)):
                poison |= dst
            else:
                poison -= dst
            if _VM_LOAD.match(op):
                lds = "_lds_" in op or re.search(r"\blds\b", args)
                hand = in_asm and not lds
                queue.append(frozenset(dst) if hand else frozenset())
                poison -= dst
                if len(queue) > 96:
                    queue.pop(0)
            d0 = ops[0] if ops else ""
            if d0 == "exec" or "saveexec" in op:
                live = op.startswith("s_or_")
            if op == "s_and_b64" and d0 == "vcc" and "exec" in ops[1:] and any(o in consts for o in ops[1:]):
                vcc = 1 if consts[[o for o in ops[1:] if o in consts][0]] else 0
            else:
                if d0.startswith("vcc") or (op.startswith("v_cmp") and "_e64" not in op):
                    vcc = None
                r = _srange(d0)
                if r:
                    for k in [k for k in consts if _srange(k)[0] <= r[1] and r[0] <= _srange(k)[1]]:
                        del consts[k]
                    if op == "s_mov_b64" and len(ops) > 1 and ops[1] in ("-1", "0"):
                        consts[d0] = int(ops[1])
            i += 1
    return sorted(problems)


def _names(regs):
    return ",".join(("a%d" % (r - 1000)) if r >= 1000 else ("v%d" % r) for r in sorted(regs))


def lint_hand_loads(asm_path):
    problems = []
    for name, body, scratch in _kernels(asm_path):
        if not any(a and _VM_LOAD.match(t.split()[0]) for t, a in body if not t.endswith(":")):
            continue
        if scratch:
            problems.append((name, "scratch %d B/lane" % scratch))
        problems += [(name, p) for p in lint_kernel(body)]
    return problems


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for p in lint_hand_loads(path):
            print("%s: %s" % p)
