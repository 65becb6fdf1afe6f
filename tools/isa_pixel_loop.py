#!/usr/bin/env python3
"""Static instruction picture of k_maniac_decode's pixel loop ("a static count of the pixel loop's ISA").

    python tools/isa_pixel_loop.py [-DFUIF_WAVES=6 ...]      # compiles fuif_amd/csrc/maniac_decode.hip to gfx950 assembly

Finds, in the DENSE configuration (k_maniac_decode<2, true>, the one the headline launch runs), the loop that holds the hand-written
symbol decoder (the inline-asm block that starts with `v_mov_b32 vN, 0x800`: fast_symbol_hw), i.e. the per-pixel loop of the
compressed slow track (encoding.cpp:386-422), and counts per basic block: scalar / vector / LDS / vector-memory / branch
instructions, waits, SGPR spill traffic (v_writelane / v_readlane with a CONSTANT lane: hipcc's spill slots) and VGPR spill
traffic (scratch_*).  Blocks are attributed with LLVM's own loop comments.  The counts are static: how often a block runs is
data; the measured dynamic mix is profiles/r3_sq_counters_6waves.txt."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SPILL_VGPRS = set()   # VGPRs hipcc keeps SGPR spill slots in: written lane by lane (v_writelane, constant lane) at >= 4 different lanes


def classify(ins, ops):
    if ins.startswith("scratch_"):
        return "vgpr_spill"
    if ins in ("v_readlane_b32", "v_writelane_b32"):
        o = [x.strip() for x in ops.split(",")]
        vreg = o[0] if ins == "v_writelane_b32" else o[1]
        if re.fullmatch(r"\d+", o[-1]) and vreg in SPILL_VGPRS:
            return "sgpr_spill"
        return "cross"
    if ins == "v_readfirstlane_b32":
        return "cross"
    if ins.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if ins.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
        return "wait"
    if ins.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if ins.startswith("s_"):
        return "salu"
    if ins.startswith(("ds_",)):
        return "lds"
    if ins.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if ins.startswith("v_"):
        return "valu"
    return "other"


def main():
    flags = [a for a in sys.argv[1:] if a != "--all"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags,
                        os.path.join(ROOT, "fuif_amd/csrc/maniac_decode.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"_ZN7fuifgpu15k_maniac_decodeILi0ELb1EEEvNS_12DecodeParamsE:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    # basic blocks: label -> (loop header it is in, is itself a header, parents)
    blocks, cur = collections.OrderedDict(), None
    for i, l in enumerate(body):
        m = re.match(r"(\.LBB0_\d+):\s*;?\s*(.*)", l)
        if m:
            cur = m.group(1)
            blocks[cur] = {"first": i, "ins": [], "header": None, "parents": [], "is_header": False, "depth": 0, "asm": []}
            tail = m.group(2)
            j = i
            comments = [tail]
            while j + 1 < len(body) and body[j + 1].strip().startswith(";") and "ASMSTART" not in body[j + 1]:
                j += 1
                comments.append(body[j].strip("; \t"))
            for c in comments:
                mm = re.search(r"in Loop: Header=(BB0_\d+) Depth=(\d+)", c)
                if mm:
                    blocks[cur]["header"] = ".L" + mm.group(1); blocks[cur]["depth"] = int(mm.group(2))
                mm = re.search(r"Parent Loop (BB0_\d+) Depth=(\d+)", c)
                if mm:
                    blocks[cur]["parents"].append(".L" + mm.group(1))
                mm = re.search(r"(?:This Inner Loop Header|This Loop Header): Depth=(\d+)", c)
                if mm:
                    blocks[cur]["is_header"] = True; blocks[cur]["depth"] = int(mm.group(1))
            continue
        if cur is None:
            continue
        s = l.strip()
        if not s or s.startswith((";", ".")) and "ASMSTART" not in s:
            continue
        if "ASMSTART" in s:
            blocks[cur]["asm"].append(body[i + 1].strip() if i + 1 < len(body) else "")
            continue
        if "ASMEND" in s or re.match(r"\d+:", s):
            continue
        parts = s.split(None, 1)
        blocks[cur]["ins"].append((parts[0], parts[1] if len(parts) > 1 else ""))
    lanes = collections.defaultdict(set)
    for b in blocks.values():
        for ins, ops in b["ins"]:
            o = [x.strip() for x in ops.split(",")]
            if ins == "v_writelane_b32" and re.fullmatch(r"\d+", o[-1]):
                lanes[o[0]].add(o[-1])
    SPILL_VGPRS.update(v for v, ls in lanes.items() if len(ls) >= 4)
    # the pixel loop: the loop that holds the hand-written symbol decoder
    hosts = [k for k, b in blocks.items() if any(re.match(r"v_mov_b32 v\d+, 0x800", a) for a in b["asm"])]
    if "--all" not in sys.argv:
        hosts = hosts[:1]   # (--all: every instantiation of the pixel loop that holds the decoder -- wide / narrow supernodes x chunk with / without the end-of-stream test)
    parents_of = {k: b["parents"] for k, b in blocks.items() if b["is_header"]}
    cols = ["salu", "valu", "cross", "lds", "vmem", "smem", "branch", "wait", "sgpr_spill", "vgpr_spill"]
    for host in hosts:
        loop = blocks[host]["header"] if not blocks[host]["is_header"] else host

        def inside(k):
            b = blocks[k]
            if k == loop or b["header"] == loop:
                return "body"
            h = k if b["is_header"] else b["header"]
            if h and loop in parents_of.get(h, []):
                return "inner loop " + h
            return None

        print("pixel loop = %s (holds fast_symbol_hw in %s), flags %s; SGPR spill slots live in %s" % (loop, host, " ".join(flags) or "(release)", " ".join(sorted(SPILL_VGPRS))))
        print("%-12s %-22s %5s " % ("block", "where", "instr") + " ".join("%10s" % c for c in cols) + "   inline asm")
        tot = collections.Counter(); n_blocks = 0
        for k, b in blocks.items():
            w = inside(k)
            if not w:
                continue
            n_blocks += 1
            c = collections.Counter(classify(i, o) for i, o in b["ins"])
            tot.update(c)
            print("%-12s %-22s %5d " % (k, w, len(b["ins"])) + " ".join("%10d" % c[x] for x in cols) + ("   " + "; ".join(a[:40] for a in b["asm"]) if b["asm"] else ""))
        print("%-12s %-22s %5d " % ("total", "%d blocks" % n_blocks, sum(tot.values())) + " ".join("%10d" % tot[x] for x in cols))
    whole = collections.Counter(classify(i, o) for b in blocks.values() for i, o in b["ins"])
    print("%-12s %-22s %5d " % ("kernel", "%d blocks" % len(blocks), sum(whole.values())) + " ".join("%10d" % whole[x] for x in cols))


if __name__ == "__main__":
    main()
