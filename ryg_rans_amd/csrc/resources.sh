#!/bin/bash
# Per-kernel register / occupancy summary from hipcc's kernel-resource-usage remarks.
cd "$(dirname "$0")"
for f in decode_wave.hip encode_wave.hip lanes.hip container_kernels.hip; do
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1
done |
python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark: (.*?): (.*?) \[-Rpass", line)
    if not m:
        m2=re.search(r"remark: Function Name: (\S+)", line)
        if m2: cur={"name":m2.group(1)}; rows.append(cur)
        continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name": cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k]=v
names=subprocess.run(["c++filt"],input="\n".join(r["name"] for r in rows),capture_output=True,text=True).stdout.split("\n")
for r,n in zip(rows,names):
    n=re.sub(r"rans_amd::\(anonymous namespace\)::","",n); n=re.sub(r"\(.*","",n); n=n.replace("void ","")
    print("%-28s vgpr=%-4s agpr=%-3s sgpr=%-4s scratch=%-4s occ=%s" % (n, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
'
