"""tools/r05/kernel_regs.py FILE.hip ... -- registers, LDS and scratch of every kernel of a translation unit (device-only assembly)."""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for src in sys.argv[1:]:
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", "-o", out, src] + os.environ.get("HF_CXXFLAGS", "").split(), check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    for n, body in re.findall(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
        g = lambda k: re.search(r'\.amdhsa_' + k + r' (\d+)', body)
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"^void hf::\(anonymous namespace\)::", "", d).split("(")[0]
        v = int(g("next_free_vgpr").group(1))
        alloc = -(-v // 8) * 8                                   # the hardware allocates in granules of 8 registers (MI355X_MICROARCH.md "Register files")
        print(f"{d[:90]:90s} vgpr {v:4d} ({min(8, 512 // max(alloc, 8))} waves/SIMD) sgpr {g('next_free_sgpr').group(1):>4} lds {g('group_segment_fixed_size').group(1):>6} scratch {g('private_segment_fixed_size').group(1)}")
