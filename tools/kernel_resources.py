#!/usr/bin/env python
"""
Build-time guard for the hand-scheduled kernels.

pixpass_wave_kernel's look-ahead loads are inline-asm global_load instructions
that the compiler does not track, waited for by hand-counted s_waitcnt
vmcnt(N).  Two things would silently break that discipline after a compiler
bump or an unrelated edit:
  * register spills (scratch reloads are VMEM operations and share vmcnt), and
  * a register allocation outside the occupancy band the kernels were tuned
    and validated in.
This script reads the AMDGPU metadata notes of every gfx950 code object
embedded in libngmix_hip.so and fails (exit status 1) when a guarded kernel
spills, uses scratch, or leaves its VGPR band.  `make` runs it after linking;
tests/test_cabi_host.py runs it again.

    python tools/kernel_resources.py [path/to/libngmix_hip.so] [--all]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# kernel-name fragment -> max VGPRs.  512 VGPRs per SIMD lane: <= 84 keeps the 6
# waves per SIMD the fused pixel kernels were tuned and validated at (they
# hold 70..78 today).
GUARDS = {
    "pixpass_wave_kernel": 84,
    # config 4's em_run: three waves per SIMD (512 / 3); at 172 registers it ran
    # 13.4 instead of 10.7 ms per 125k stamps
    "em_wave_kernel<64, 16, 0, 1, 1>": 168,
}
# register-heavy on purpose (the stamp or the accumulators live in VGPRs):
# only memory spills are an error
NO_SPILL_ONLY = ("admom_grid_kernel", "em_wave_kernel", "wsums_wave_kernel",
                 "lm_eval_kernel", "lm_eval_fd_kernel")


def code_objects(lib):
    """yield the gfx950 ELF images of every offload bundle in .hip_fatbin"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"),
                        "--dump-section", ".hip_fatbin=" + fat, lib,
                        os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + len(MAGIC))


def kernel_table(lib):
    """{demangled kernel name: metadata dict} for every kernel in the library"""
    table = {}
    for image in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name],
                                   capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            blk = ".agpr_count:" + blk
            fields = dict(re.findall(r"\.(\w+):\s+('?[^\n']+'?)", blk))
            name = fields.get("name", "").strip("'")
            if not name:
                continue
            table[name] = {k: int(v) for k, v in fields.items()
                           if re.fullmatch(r"\d+", v.strip())}
    names = list(table)
    dem = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True, check=True).stdout.splitlines()
    return {d: table[n] for n, d in zip(names, dem)}


def check(lib, verbose=False):
    table = kernel_table(lib)
    problems = []
    seen = set()
    for name, md in sorted(table.items()):
        # SGPR spills go to VGPR lanes (v_writelane), not to memory: only VGPR
        # spills and a private segment mean scratch traffic
        vspill = md.get("vgpr_spill_count", 0)
        sspill = md.get("sgpr_spill_count", 0)
        scratch = md.get("private_segment_fixed_size", 0)
        vgpr = md.get("vgpr_count", 0)
        guard = next((k for k in GUARDS if k in name), None)
        nospill = next((k for k in NO_SPILL_ONLY if k in name), None)
        if verbose:
            print("%-78s vgpr %3d sgpr %3d vspill %d sspill %d scratch %d lds %d" % (
                name[:78], vgpr, md.get("sgpr_count", 0), vspill, sspill, scratch,
                md.get("group_segment_fixed_size", 0)))
        if guard:
            seen.add(guard)
            if vspill or scratch:
                problems.append("%s: %d spilled VGPRs, %d B scratch (hand-counted "
                                "vmcnt would be wrong)" % (name, vspill, scratch))
            if vgpr > GUARDS[guard]:
                problems.append("%s: %d VGPRs > %d (outside the validated occupancy "
                                "band)" % (name, vgpr, GUARDS[guard]))
        elif nospill:
            seen.add(nospill)
            # (VGPRs parked in AGPRs count as spilled but cost one move each and
            # no memory: scratch is what must not happen)
            if scratch:
                problems.append("%s: %d B scratch (%d spilled VGPRs)" % (
                    name, scratch, vspill))
    for k in list(GUARDS) + list(NO_SPILL_ONLY):
        if k not in seen:
            problems.append("guarded kernel %s not found in %s" % (k, lib))
    return table, problems


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    args = [a for a in argv if not a.startswith("--")]
    lib = args[0] if args else os.path.join(here, "..", "ngmix_amd", "libngmix_hip.so")
    _, problems = check(lib, verbose="--all" in argv)
    for p in problems:
        sys.stderr.write("kernel_resources: " + p + "\n")
    if not problems:
        print("kernel_resources: %s ok (no scratch / VGPR spills in the guarded kernels)"
              % os.path.basename(lib))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
