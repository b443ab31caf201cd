"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage output (make -C csrc asm) per render kernel."""
import re, sys, subprocess
t = open(sys.argv[1] if len(sys.argv) > 1 else 'build/resource_usage.txt').read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]
    if 'render_b' not in name and 'resolve' not in name:
        continue
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('(rtw::KArgs)', '')
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, '?'])[1]
    print(f"{dn:52s} VGPR {g('VGPRs'):>3} SGPR {g('SGPRs'):>3} scratch {g('ScratchSize .bytes/lane.'):>3} waves/SIMD {g('Occupancy .waves/SIMD.')}")
