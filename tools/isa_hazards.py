#!/usr/bin/env python
"""
ISA-level hazard check of the shipped gfx950 code objects (round-2 verdict,
item 3: "hand-scheduled asm is only as safe as the tests that run on it").

The kernels issue look-ahead loads from inline asm (the compiler does not know
them) and wait for them with hand-counted `s_waitcnt vmcnt(N)`; they move
wave-uniform values to SGPRs with `v_readfirstlane` in inline asm, which the
compiler's hazard recognizer does not look into.  Both failure modes are
silent (a stale register now and then), so this script proves their absence
on the machine code that ships, for EVERY function of every code object:

  A. vmcnt discipline.  A forward dataflow over the control-flow graph tracks,
     per VGPR/AGPR, the youngest memory load that may still be in flight into
     it as k = the number of VMEM operations issued after it (on gfx9 loads
     and stores share vmcnt and retire in issue order -- stores only got their
     own counter with gfx10 -- which is the model of LLVM's SIInsertWaitcnts
     and what the compiler's own code relies on: counting loads alone flags
     compiler-made `load; store; s_waitcnt vmcnt(1)` sequences).
     `s_waitcnt vmcnt(N)` retires every register with k >= N.  Any instruction
     that reads or writes a register whose load may be in flight is an error
     -- a v_mov copying a look-ahead register, a spill, a wait count that is
     one too high.  At joins the state is the per-register minimum of k
     (in flight on any path = in flight).
  B. VALU write of a VGPR -> v_readlane / v_readfirstlane of it: 1 wait state
     (gfx90a+: LLVM GCNHazardRecognizer VALUWriteVGPRReadlaneRead).
  C. VALU write of a VGPR -> DPP read of it: 2 wait states.
  D. VALU write of an SGPR (v_readfirstlane, v_cmp, carry-out) -> VMEM
     instruction reading that SGPR: 5 wait states.
  E. VALU write of an SGPR -> v_readlane / v_writelane lane select: 4.

B-E walk every CFG path backwards for the required number of wait states
(`s_nop N` = N + 1, any other instruction = 1).  The compiler keeps these rules
for its own instructions; a report here is an inline-asm instruction it could
not see.

    python tools/isa_hazards.py [libngmix_hip.so | file.co | file.o ...] [-v]
                                [--only=name-fragment ...]

`make` runs it on the freshly linked library (a failure deletes the library);
tests/test_cabi_host.py runs it on a deliberately broken kernel
(tests/helpers/hazard_cases.hip) and expects every case to be reported.
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402

LLVM = kr.LLVM
KMAX = 63                       # vmcnt is six bits on gfx9

_REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")
_DPP = ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "wave_shl:", "wave_shr:",
        "wave_rol:", "wave_ror:", "row_mirror", "row_half_mirror", "row_bcast:",
        "row_newbcast:", "row_share:", "row_xmask:")


def regs_of(operand):
    """[(file, index)] of every register named in one operand string"""
    out = []
    for m in _REG.finditer(operand):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out.extend((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    if re.search(r"\bvcc(_lo)?\b", operand):
        out.append(("s", 106))
    if re.search(r"\bvcc(_hi)?\b", operand) and "vcc_lo" not in operand:
        out.append(("s", 107))
    return out


class Insn(object):
    __slots__ = ("addr", "text", "op", "operands", "is_vmem", "is_vmem_load", "is_valu",
                 "touched", "vdst", "sdst", "vsrc", "ssrc", "is_dpp", "nop_states")

    def __init__(self, addr, text):
        self.addr, self.text = addr, text
        parts = text.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        self.operands = [o.strip() for o in rest.split(",")] if rest else []
        op = self.op
        self.is_vmem = op.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_"))
        self.is_vmem_load = self.is_vmem and ("_load" in op or (
            "_atomic" in op and any(x in rest for x in (" glc", " sc0"))))
        self.is_valu = op.startswith("v_")
        self.is_dpp = self.is_valu and ("_dpp" in op or any(d in rest for d in _DPP))
        self.nop_states = 1
        if op == "s_nop":
            try:
                self.nop_states = int(self.operands[0], 0) + 1
            except (ValueError, IndexError):
                pass
        allregs = [regs_of(o) for o in self.operands]
        self.touched = set(r for rs in allregs for r in rs if r[0] in "va")
        first = allregs[0] if allregs else []
        dst = list(first)
        if op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")) and len(allregs) > 1:
            dst = dst + allregs[1]
        if self.is_vmem and not self.is_vmem_load:
            dst = []                      # stores / non-returning atomics
        if op.startswith(("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop",
                          "s_endpgm", "s_barrier", "s_setprio", "s_sleep", "s_sendmsg",
                          "v_cmpx", "ds_write", "ds_store", "s_store", "s_dcache",
                          "buffer_wbl2", "buffer_inv", "s_setreg", "s_set_gpr_idx",
                          "v_nop", "s_icache")):
            dst = []
        self.vdst = set(r for r in dst if r[0] in "va")
        self.sdst = set(r for r in dst if r[0] == "s")
        # carry-out / second scalar destination of VOP3 forms (v_add_co_u32 v, s[..], ..;
        # v_mad_u64_u32 v[..], s[..], ..; v_div_scale)
        if self.is_valu and len(allregs) > 1 and op.startswith(
                ("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co",
                 "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")):
            self.sdst |= set(r for r in allregs[1] if r[0] == "s")
        src = set(r for rs in (allregs[1:] if dst else allregs) for r in rs)
        if op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")):
            src |= set(first)
        self.vsrc = set(r for r in src if r[0] in "va")
        self.ssrc = set(r for r in src if r[0] == "s")


class Block(object):
    def __init__(self, name):
        self.name = name
        self.insns = []
        self.succ = []
        self.pred = []


def disassemble(path):
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950",
                          "--symbolize-operands", path],
                         capture_output=True, text=True, check=True).stdout
    return out


_HDR = re.compile(r"^([0-9a-f]{8,16}) <([^>]+)>:\s*$")


def functions(disasm, only=None):
    """yield (name, [Block]) per function (those whose name contains one of
    `only`, if given); blocks linked into a CFG"""
    cur_name, raw = None, []          # raw: list of ("label", name) / ("insn", Insn)
    skip = False
    def finish():
        if cur_name is not None and raw:
            yield_fn.append((cur_name, raw[:]))
    yield_fn = []
    for line in disasm.splitlines():
        m = _HDR.match(line)
        if m:
            label = m.group(2)
            if re.fullmatch(r"L\d+", label):
                raw.append(("label", label))
            else:
                finish()
                cur_name, raw = label, []
                skip = bool(only) and not any(o in label for o in only)
                if skip:
                    cur_name = None
            continue
        if skip or not line.startswith("\t"):
            continue
        text, _, comment = line.strip().partition("//")
        text = text.strip()
        if not text or text.startswith("."):
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        addr = int(am.group(1), 16) if am else 0
        raw.append(("insn", Insn(addr, text)))
    finish()
    for name, items in yield_fn:
        blocks, byname = [], {}
        cur = Block("entry")
        blocks.append(cur)
        for kind, v in items:
            if kind == "label":
                nb = Block(v)
                byname[v] = nb
                blocks.append(nb)
                cur = nb
            else:
                cur.insns.append(v)
                if v.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                    nb = Block(None)
                    blocks.append(nb)
                    cur = nb
        for i, b in enumerate(blocks):
            last = b.insns[-1] if b.insns else None
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            fall = True
            if last is not None:
                if last.op.startswith("s_cbranch") or last.op == "s_branch":
                    tgt = byname.get(last.operands[0]) if last.operands else None
                    if tgt is not None:
                        b.succ.append(tgt)
                    fall = last.op != "s_branch"
                elif last.op.startswith(("s_endpgm", "s_setpc")):
                    fall = False
            if fall and nxt is not None:
                b.succ.append(nxt)
        for b in blocks:
            for s in b.succ:
                s.pred.append(b)
        yield name, blocks


# ------------------------------------------------------------------ check A

def _join(states):
    out = {}
    for st in states:
        for r, k in st.items():
            if r not in out or k < out[r]:
                out[r] = k
    return out


def _vmcnt(insn):
    if insn.op.startswith("s_swappc"):
        # a call: the callee waits for every counter before it returns
        # (s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0) at entry, AMDGPU calling convention)
        return 0
    if insn.op != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", insn.text)
    if m:
        return int(m.group(1))
    # a raw immediate: vmcnt = bits [3:0] | bits [15:14] << 4
    if insn.operands and re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", insn.operands[0]):
        imm = int(insn.operands[0], 0)
        return (imm & 0xF) | ((imm >> 14) & 0x3) << 4
    return None


def check_vmcnt(name, blocks, problems, stats):
    entry = {id(b): None for b in blocks}
    exit_ = {id(b): None for b in blocks}
    entry[id(blocks[0])] = {}
    work = [blocks[0]]
    reported = set()

    def transfer(b, st, report):
        st = dict(st)
        for ins in b.insns:
            n = _vmcnt(ins)
            if n is not None:
                for r in [r for r, k in st.items() if k >= n]:
                    del st[r]
                continue
            if st:
                hit = [r for r in ins.touched if r in st]
                # (the load's own destination may be re-targeted; its address
                # and data registers may not be in flight)
                if ins.is_vmem_load:
                    hit = [r for r in hit if r not in ins.vdst or r in ins.vsrc]
                if hit and report and (ins.addr, name) not in reported:
                    reported.add((ins.addr, name))
                    problems.append(
                        "%s: %#x `%s` touches %s while a load into it may be in flight "
                        "(vmcnt <= %d would be needed before it)" % (
                            name, ins.addr, ins.text,
                            ", ".join("%s%d" % r for r in sorted(hit)),
                            min(st[r] for r in hit)))
            if ins.is_vmem:
                for r in list(st):
                    st[r] = min(st[r] + 1, KMAX)
                if ins.is_vmem_load:
                    stats["loads"] += 1
                    for r in ins.vdst:
                        st[r] = 0
        return st

    guard = 0
    while work:
        guard += 1
        if guard > 200000:
            problems.append("%s: vmcnt dataflow did not converge" % name)
            return
        b = work.pop()
        st_in = entry[id(b)]
        st_out = transfer(b, st_in, False)
        if exit_[id(b)] == st_out:
            continue
        exit_[id(b)] = st_out
        for s in b.succ:
            preds = [exit_[id(p)] for p in s.pred if exit_[id(p)] is not None]
            new = _join(preds)
            if entry[id(s)] != new:
                entry[id(s)] = new
                work.append(s)
            elif exit_[id(s)] is None:
                work.append(s)
    for b in blocks:
        if entry[id(b)] is not None:
            transfer(b, entry[id(b)], True)


# -------------------------------------------------------------- checks B - E

def _lookback(blocks_index, b, idx, need, hit):
    """does any path reach (b, idx) with an instruction satisfying hit() fewer
    than `need` wait states before it?  Returns the offending Insn or None."""
    stack = [(b, idx - 1, 0)]
    seen = set()
    while stack:
        blk, i, states = stack.pop()
        while i >= 0:
            ins = blk.insns[i]
            if hit(ins):
                return ins
            states += ins.nop_states
            if states >= need:
                break
            i -= 1
        else:
            for p in blk.pred:
                key = (id(p), states)
                if key not in seen:
                    seen.add(key)
                    stack.append((p, len(p.insns) - 1, states))
    return None


def check_wait_states(name, blocks, problems, stats):
    for b in blocks:
        for i, ins in enumerate(b.insns):
            rules = []
            if ins.op in ("v_readfirstlane_b32", "v_readlane_b32"):
                src = set(regs_of(ins.operands[1])) if len(ins.operands) > 1 else set()
                vs = set(r for r in src if r[0] in "va")
                stats["readlanes"] += 1
                rules.append((1, lambda p, vs=vs: p.is_valu and p.vdst & vs,
                              "VALU write of a VGPR -> v_readlane/v_readfirstlane (1)"))
            if ins.op in ("v_readlane_b32", "v_writelane_b32") and len(ins.operands) > 2:
                sel = set(r for r in regs_of(ins.operands[2]) if r[0] == "s")
                if sel:
                    rules.append((4, lambda p, sel=sel: p.is_valu and p.sdst & sel,
                                  "VALU write of an SGPR -> lane select (4)"))
            if ins.is_dpp:
                vs = ins.vsrc
                stats["dpp"] += 1
                rules.append((2, lambda p, vs=vs: p.is_valu and p.vdst & vs,
                              "VALU write of a VGPR -> DPP read (2)"))
            if ins.is_vmem:
                ss = set(r for o in ins.operands for r in regs_of(o) if r[0] == "s")
                if ss:
                    rules.append((5, lambda p, ss=ss: p.is_valu and p.sdst & ss,
                                  "VALU write of an SGPR -> VMEM read of it (5)"))
            for need, hit, what in rules:
                bad = _lookback(None, b, i, need, hit)
                if bad is not None:
                    problems.append("%s: %#x `%s` too close after %#x `%s`: %s" % (
                        name, ins.addr, ins.text, bad.addr, bad.text, what))


def check_file(path, problems, stats, verbose=False, only=None):
    dis = disassemble(path)
    for name, blocks in functions(dis, only):
        stats["functions"] += 1
        n0 = len(problems)
        check_vmcnt(name, blocks, problems, stats)
        check_wait_states(name, blocks, problems, stats)
        if verbose:
            print("  %-90s %5d insns %s" % (name[:90], sum(len(b.insns) for b in blocks),
                                           "ok" if len(problems) == n0 else "PROBLEMS"))


def images_of(path):
    """code-object files to disassemble for `path` (a host library / object
    with a .hip_fatbin section, or a bare gfx950 code object)"""
    with open(path, "rb") as f:
        head = f.read(20)
    # e_machine 0xE0 = EM_AMDGPU
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":
        yield path, None
        return
    sections = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", path],
                              capture_output=True, text=True).stdout
    if ".hip_fatbin" not in sections:
        return                              # host-only object: no device code
    for image in kr.code_objects(path):
        f = tempfile.NamedTemporaryFile(suffix=".co", delete=False)
        f.write(image)
        f.close()
        yield f.name, f.name


def check(paths, verbose=False, only=None):
    problems = []
    stats = {"functions": 0, "loads": 0, "readlanes": 0, "dpp": 0}
    for path in paths:
        for co, tmp in images_of(path):
            try:
                check_file(co, problems, stats, verbose, only)
            finally:
                if tmp:
                    os.unlink(tmp)
    return problems, stats


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    verbose = "-v" in argv
    paths = [a for a in argv if not a.startswith("-")]
    only = [a.split("=", 1)[1] for a in argv if a.startswith("--only=")] or None
    if not paths:
        paths = [os.path.join(here, "..", "ngmix_amd", "libngmix_hip.so")]
    problems, stats = check(paths, verbose, only)
    for p in problems:
        sys.stderr.write("isa_hazards: " + p + "\n")
    if not problems:
        print("isa_hazards: %s ok (%d functions: %d loads followed to their waits, %d "
              "readlanes, %d DPP reads checked)" % (
                  ", ".join(os.path.basename(p) for p in paths), stats["functions"],
                  stats["loads"], stats["readlanes"], stats["dpp"]))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
