# check: in every x2/x4 kernel, no v_accvgpr_write within 2 wait states in front of an MFMA that reads that AGPR
from tests import codeobj
import tempfile, subprocess, re, sys
d=tempfile.mkdtemp()
lib=sys.argv[1] if len(sys.argv)>1 else 'flashattention.c_amd/libflashattn_amd.so'
ks=codeobj.kernels_of(lib, d)
for k in ks.values():
    if 'fa_fwd_bf16_x' not in k.name: continue
    out=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d',k.code_object],capture_output=True,text=True).stdout.splitlines()
    start=[i for i,l in enumerate(out) if k.mangled in l and l.endswith('>:')][0]
    end=start+1
    while end<len(out) and not re.match(r'^[0-9a-f]+ <',out[end]): end+=1
    body=[re.sub(r'\s+//.*','',x).strip() for x in out[start+1:end]]
    n=0
    for i,l in enumerate(body):
        if l.startswith('v_mfma'):
            rng=[(int(a),int(b)) for a,b in re.findall(r'a\[(\d+):(\d+)\]',l)]
            ws=0
            for j in range(i-1,max(0,i-6),-1):
                pj=body[j]
                if pj.startswith('s_nop'): ws+=int(pj.split()[1])+1
                elif pj.startswith('v_mfma'): break
                else:
                    w=re.match(r'v_accvgpr_write_b32 a(\d+),',pj)
                    if w and any(a<=int(w.group(1))<=b for a,b in rng) and ws<2:
                        n+=1; break
                    ws+=1
                if ws>=2: break
    print(n, k.vgprs, k.agprs, k.name[:100])
