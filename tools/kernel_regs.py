"""Register / LDS / scratch use of the library's kernels, read from the code objects inside libstylemesh_hip.so (no GPU).
Usage: kernel_regs.py [substring of the kernel name = conv3x3_split] [lib]"""
import os, re, subprocess, sys, tempfile
pat = sys.argv[1] if len(sys.argv) > 1 else "conv3x3_split"
lib = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "stylemesh_amd", "libstylemesh_hip.so"))
llvm = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, "lib.so")
    os.symlink(lib, tmp)
    subprocess.run([llvm + "llvm-objdump", "--offloading", tmp], check=True, capture_output=True, cwd=d)
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([llvm + "llvm-readelf", "--notes", os.path.join(d, f)], capture_output=True, text=True).stdout
        for e in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s*(\S+)", e).group(1)
            if pat not in name:
                continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            g = lambda k: re.search(r"\." + k + r":\s*(\d+)", e).group(1)
            print(f"vgpr {g('vgpr_count'):>3s} agpr {e.split()[0]:>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} "
                  f"scratch {g('private_segment_fixed_size'):>4s}  {dem[:150]}")
