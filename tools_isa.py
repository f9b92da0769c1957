"""Register and spill summary of the device kernels: python3 tools_isa.py f64|f32 [extra compiler flags]
Compiles the translation units to gfx950 assembly under /tmp and prints, per kernel, the VGPR /
SGPR / scratch figures and the counts of the instructions that indicate register pressure."""
import re, subprocess, sys

prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
contract = "off" if prec == "f64" else "fast"
extra = [] if prec == "f64" else ["-fno-slp-vectorize"]  # as build.sh compiles the f32 kernels
s = ""
for unit, more in (("render", []), ("bounce", ["-mllvm", "-disable-machine-licm"])):
    asm = "/tmp/%s_%s.s" % (unit, prec)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=" + contract,
                    "--cuda-device-only", "-S", "-o", asm, "ray-tracer_amd/csrc/%s_%s.hip" % (unit, prec)] + extra + more + sys.argv[2:],
                   check=True, stderr=subprocess.DEVNULL)
    s += open(asm).read()
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    g = lambda k: re.search(k + r"\s+(\S+)", m.group(2)).group(1)
    meta[m.group(1)] = (g(".amdhsa_next_free_vgpr"), g(".amdhsa_next_free_sgpr"), g(".amdhsa_accum_offset"),
                        g(".amdhsa_private_segment_fixed_size"))
for name, (v, sg, acc, scr) in meta.items():
    i = s.index("\n" + name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j]
    cnt = lambda pat: len(re.findall(pat, body))
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn)
    print("%-48s vgpr %3s accum_off %3s sgpr %3s scratch %4s | lines %5d valu %5d readlane %3d writelane %3d "
          "scratch_ld/st %d/%d s_load %d waitcnt %d" % (dn[-48:], v, acc, sg, scr, body.count("\n"), cnt(r"\n\s+v_"),
          cnt("v_readlane"), cnt("v_writelane"), cnt("scratch_load"), cnt("scratch_store"), cnt(r"\ss_load"), cnt("s_waitcnt")))
