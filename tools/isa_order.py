"""Instruction ORDER of a kernel's ISA, compressed: which memory operations the compiler finally placed where relative to the MFMAs and barriers.
    python tools/isa_order.py yolo_v3_amd/csrc/conv_igemm_f32.hip [kernel-name-substring]
One line per kernel (in layout order of the assembly; loops appear once):  L global load, D global_load_lds (DMA), S global store, R ds_read, W ds_write,
M MFMA, | s_barrier, (vN) / (kN) s_waitcnt vmcnt(N) / lgkmcnt(N); runs are counted (M32 = 32 MFMAs in a row).
Round 5 found this way: in conv_igemm_f32_kernel's 1x1 / Winograd instantiations the scheduler had sunk the chunk prefetch to the end of the iteration
('|(v4)W1...(v0)W1 M23 L1 M4 L3' instead of '|L6 (v11)W1..(v6)W1 M32'), profiles/r05z6_f32_pinned_prefetch_ab.txt."""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
                "-I" + os.path.join(root, "yolo_v3_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out] + sys.argv[3:],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if flt not in name or "v_mfma" not in body and "global_load" not in body:
        continue
    seq = []
    for l in body.split("\n"):
        t = l.strip().split(" ")[0] if l.strip() else ""
        if t.startswith("global_load") and " lds" in l: seq.append("D")
        elif t.startswith(("global_load", "buffer_load")): seq.append("L")
        elif t.startswith(("global_store", "buffer_store")): seq.append("S")
        elif t.startswith("ds_write"): seq.append("W")
        elif t.startswith("ds_read"): seq.append("R")
        elif t.startswith("v_mfma"): seq.append("M")
        elif t == "s_barrier": seq.append("|")
        elif t.startswith("s_waitcnt"):
            v, k = re.search(r"vmcnt\((\d+)\)", l), re.search(r"lgkmcnt\((\d+)\)", l)
            if v or k: seq.append(("v" + v.group(1) if v else "") + ("k" + k.group(1) if k else ""))
    txt = "".join(x if len(x) == 1 else "(" + x + ")" for x in seq)
    for ch in "MLWRSD":
        txt = re.sub(ch + "+", lambda m: ch + str(len(m.group(0))), txt)
    print(name[:160]); print("   ", txt[:int(os.environ.get("WIDTH", "600"))])
