import sys,re,subprocess,struct,tempfile,os
LIB=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else "icp_step_fast"
LLVM="/opt/rocm/lib/llvm/bin"; MAGIC=b"__CLANG_OFFLOAD_BUNDLE__"
d=tempfile.mkdtemp()
subprocess.check_call([f"{LLVM}/llvm-objcopy",f"--dump-section=.hip_fatbin={d}/fat.bin",LIB,f"{d}/x.so"])
blob=open(f"{d}/fat.bin","rb").read()
k=0
for m in re.finditer(MAGIC,blob):
    base=m.start(); (cnt,)=struct.unpack_from("<Q",blob,base+len(MAGIC)); at=base+len(MAGIC)+8
    for _ in range(cnt):
        off,size,tlen=struct.unpack_from("<QQQ",blob,at); triple=blob[at+24:at+24+tlen].decode(); at+=24+tlen
        if "gfx950" in triple and size:
            f=f"{d}/co{k}.elf"; k+=1; open(f,"wb").write(blob[base+off:base+off+size])
            txt=subprocess.run([f"{LLVM}/llvm-readelf","--notes",f],capture_output=True,text=True).stdout
            cur={}
            for line in txt.splitlines():
                line=line.strip()
                mm=re.match(r"[-\s]*\.(\w+):\s*(.*)",line)
                if not mm: continue
                key,val=mm.groups()
                if key in("vgpr_count","vgpr_spill_count","sgpr_spill_count","private_segment_fixed_size","sgpr_count","agpr_count","group_segment_fixed_size"): cur[key]=val
                if key=="name" or key=="symbol":
                    pass
                if key=="name" and pat in val and not val.startswith("_ZN") is None:
                    cur["name"]=val
                if key=="wavefront_size":
                    if pat in cur.get("name",""): print(cur)
                    cur={}
