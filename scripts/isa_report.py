#!/usr/bin/env python
"""What hipcc made of a kernel, without a GPU: registers / scratch / LDS, the instruction mix of its largest loop, and the ORDER of
the memory, matrix and barrier events in every block that holds matrix instructions.

Round 3 found two real problems this way: one VALU add per LDS read in the edge kernels (the table layout made the
compiler rebuild addresses), and node_proj_kernel's prefetch sunk next to its ds_writes (load -> wait -> write -> barrier -> MFMA,
nothing overlapped).  Usage:
    python scripts/isa_report.py cbgbench_amd/csrc/node_mfma.hip node_proj_kernel [-DFLAG ...]
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def compile_to_isa(source, extra):
    out = os.path.join(tempfile.mkdtemp(prefix="isa_report_"), "k.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-o", out, source] + extra
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return open(out).read()


def kernels(text, needle):
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        if needle in m.group(1):
            d = m.group(2)
            grab = lambda k: int(re.search(k + r" (\d+)", d).group(1))
            yield m.group(1), {"vgpr": grab("next_free_vgpr"), "scratch_bytes": grab("private_segment_fixed_size"),
                               "lds_bytes": grab("group_segment_fixed_size")}


def body_of(text, name):
    start = re.search(r"^" + re.escape(name) + r":", text, flags=re.M).start()
    body = text[start:]
    return body[:body.index("s_endpgm")]


def instructions(block):
    return [l.strip() for l in block.splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]


def largest_loop(body):
    """(first, last) instruction index of the largest backward branch"""
    lines, labels = [], {}
    for l in body.splitlines():
        l = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(lines)
        elif l and not l.startswith((";", ".")) and not l.endswith(":"):
            lines.append(l)
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"^s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], i)
    return lines, best


def event_order(block):
    out = []

    def add(tag):
        if out and out[-1][0] == tag:
            out[-1][1] += 1
        else:
            out.append([tag, 1])
    for l in block.splitlines():
        l = l.strip()
        if not l:
            continue
        k = l.split()[0]
        if "sched_barrier" in l: add("<sched_barrier>")
        elif k.startswith("global_load") or k.startswith("buffer_load"): add("GLOAD")
        elif k.startswith("global_store") or k.startswith("buffer_store"): add("GSTORE")
        elif k.startswith("global_atomic"): add("GATOMIC")
        elif k.startswith("scratch_"): add("SCRATCH")
        elif k.startswith("ds_write") or k.startswith("ds_add") or k.startswith("ds_cmpst"): add("LDSW")
        elif k.startswith("ds_read"): add("LDSR")
        elif "mfma" in k: add("MFMA")
        elif k == "s_barrier": add("BARRIER")
        elif k == "s_waitcnt" and "vmcnt" in l: add("wait " + l.split(None, 1)[1].replace(" ", ""))
    return " | ".join(t if n == 1 else f"{t} x{n}" for t, n in out)


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    source, needle, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    text = compile_to_isa(source, extra)
    for name, res in kernels(text, needle):
        print(f"== {name}\n   {res}")
        body = body_of(text, name)
        lines, loop = largest_loop(body)
        if loop:
            ops = collections.Counter(l.split()[0] for l in lines[loop[0]:loop[1]])
            valu = sum(v for k, v in ops.items() if k.startswith("v_") and "mfma" not in k)
            print(f"   largest loop: {loop[1] - loop[0]} instructions; VALU {valu}, MFMA {sum(v for k, v in ops.items() if 'mfma' in k)}, "
                  f"LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'scratch_')))}, "
                  f"SALU {sum(v for k, v in ops.items() if k.startswith('s_'))}")
            print("   top:", ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
        blocks = re.split(r"^(\.LBB\d+_\d+):", body, flags=re.M)
        for j in range(1, len(blocks), 2):
            b = blocks[j + 1]
            if b.count("v_mfma") >= 4 or (b.count("global_load") >= 4 and b.count("ds_write") >= 1):
                print(f"   {blocks[j]}: {event_order(blocks[j + 1])[:1400]}")


if __name__ == "__main__":
    main()
