import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-I/root/repo/include","-I/root/repo/nbed_amd/csrc","-c",src,"-o","/tmp/_ru.o","-Rpass-analysis=kernel-resource-usage"],capture_output=True,text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":",1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":",1); rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
    print(f"{name:90s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} SGPR {v.get('SGPRs')} occ {v.get('Occupancy [waves/SIMD]')} scratch {v.get('ScratchSize [bytes/lane]')} LDS {v.get('LDS Size [bytes/block]')}")
