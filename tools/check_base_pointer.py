#!/usr/bin/env python3
"""Scan a built object (or libzkt_hip.so) for the base-pointer hazard of DESIGN.md §5 "A compiler limit".

A device function with 64-byte aligned locals realigns its stack and keeps the incoming stack pointer in s34 (prologue `s_mov_b32 s34, s32`); it restores
the caller's s34 on the way out.  This toolchain's inter-procedural register allocation does not treat s34 as live across the calls such a function makes,
and lets callees that do NOT set up a base pointer use s34 as an ordinary scratch register.  A caller whose base pointer was trampled addresses its frame
through garbage after the call: a memory fault inside the scratch aperture, seen as a silent abort of the process.

usage: tools/check_base_pointer.py build/obj/zkt_pairing.o [more objects]      exit status 1 if a hazard is found
The call graph is recovered from the linked code object's disassembly (s_getpc_b64 / s_add_u32 / s_addc_u32 / s_swappc_b64 sequences)."""
import collections, os, re, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path):
    d = tempfile.mkdtemp()
    dst = os.path.join(d, "in.o")
    subprocess.check_call(["cp", path, dst])
    subprocess.run([OBJDUMP, "--offloading", dst], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if "gfx950" in f]


def writes_s34(op, args):
    if not args or op.startswith(("s_cmp", "s_bitcmp", "s_setpc", "s_swappc", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep")):
        return False
    if not (op.startswith("s_") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_cmp")):
        return False
    dst = args.split(",")[0].strip()
    if dst == "s34": return True
    m = re.match(r"s\[(\d+):(\d+)\]$", dst)
    return bool(m) and int(m.group(1)) <= 34 <= int(m.group(2))


def scan(co):
    funcs, starts = collections.OrderedDict(), {}
    cur = None
    for line in subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
        if m:
            cur = m.group(2); funcs[cur] = []; starts[int(m.group(1), 16)] = cur
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur: funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    bp, direct, calls = {}, {}, collections.defaultdict(set)
    for f, ins in funcs.items():
        bp[f] = any(op == "s_mov_b32" and a.replace(" ", "") == "s34,s32" for _, op, a in ins)
        # a callee that treats s34 as callee-saved (what the ABI says, and what the compiler emits when the inter-procedural register allocation is off:
        # -mllvm -enable-ipra=0) parks it in a VGPR lane before its first write and brings it back before it returns: `v_writelane_b32 vN, s34, L` ... `v_readlane_b32 s34, vN, L`
        saved = None
        for _, op, a in ins:
            if op == "v_writelane_b32" and re.match(r"v\d+,\s*s34,", a): saved = a.split(",")[0].strip() + "," + a.split(",")[2].strip(); break
            if writes_s34(op, a): break
        restored = saved is not None and any(op == "v_readlane_b32" and a.replace(" ", "") == "s34," + saved for _, op, a in ins)
        direct[f] = (not bp[f]) and not restored and any(writes_s34(op, a) for _, op, a in ins)
        for i, (addr, op, a) in enumerate(ins):
            if op != "s_getpc_b64" or i + 2 >= len(ins): continue
            m = re.match(r"s\[(\d+):(\d+)\]", a)
            lo, hi = ins[i + 1], ins[i + 2]
            if not (m and lo[1] == "s_add_u32" and hi[1] == "s_addc_u32"): continue
            try:
                off_lo = int(lo[2].split(",")[-1].strip(), 0) & 0xFFFFFFFF
                off_hi = int(hi[2].split(",")[-1].strip(), 0) & 0xFFFFFFFF
            except ValueError:
                continue
            off = (off_hi << 32) | off_lo
            if off >= 1 << 63: off -= 1 << 64
            tgt = addr + 4 + off
            if tgt in starts: calls[f].add(starts[tgt])
    clob = {}
    def clobbers(f, seen=()):
        if f in clob: return clob[f]
        if bp.get(f): clob[f] = False; return False
        if f in seen: return False
        r = direct.get(f, False) or any(clobbers(g, seen + (f,)) for g in calls[f])
        clob[f] = r
        return r
    bad = []
    for f in funcs:
        if bp[f]:
            for g in calls[f]:
                if clobbers(g): bad.append((f, g))
    return bad, sum(bp.values()), sum(len(v) for v in calls.values())


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:110]


def main():
    rc = 0
    for path in sys.argv[1:]:
        for co in code_objects(path):
            bad, nbp, ncalls = scan(co)
            print(f"{os.path.basename(path)}: {nbp} functions keep a base pointer, {ncalls} call edges, {len(bad)} hazards")
            for f, g in bad:
                rc = 1
                print(f"  HAZARD  {demangle(f)}\n      calls {demangle(g)}  (s34 written without a base-pointer frame)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
