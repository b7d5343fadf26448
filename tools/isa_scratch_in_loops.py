#!/usr/bin/env python3
"""Where a kernel's scratch instructions stand: inside which loop (the innermost backward branch around them), or outside
every loop.  Input: the disassembly of a gfx950 code object and a piece of the kernel's mangled name.

    B=/opt/rocm/lib/llvm/bin
    $B/llvm-objcopy -O binary --only-section=.hip_fatbin sybil_amd/csrc/kernels_packed_0.o fat.bin
    $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=k.co
    $B/llvm-objdump -d --mcpu=gfx950 k.co > k.s
    python tools/isa_scratch_in_loops.py k.s k_emit_packedILi0ELi1ELi1E
"""
import re,sys
s=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
starts=[(i,l) for i,l in enumerate(s) if re.match(r'^[0-9a-f]{16} <',l)]
for idx,(i,l) in enumerate(starts):
    if pat in l:
        j=starts[idx+1][0] if idx+1<len(starts) else len(s)
        body=s[i:j]
        base=int(re.match(r'^([0-9a-f]{16})',l).group(1),16)
        loops=[]
        for x in body:
            m=re.search(r's_cbranch\w*\s+\d+\s+//\s+([0-9A-F]+):.*<[^+]+\+0x([0-9a-f]+)>',x) or re.search(r's_branch\s+\d+\s+//\s+([0-9A-F]+):.*<[^+]+\+0x([0-9a-f]+)>',x)
            if m:
                cur=int(m.group(1),16)-base; tgt=int(m.group(2),16)
                if tgt<cur: loops.append((tgt,cur))
        print(l.split('<')[1][:80])
        print(" instructions",len(body),"backward branches",len(loops))
        for k,x in enumerate(body):
            if 'scratch_' in x:
                m=re.search(r'//\s+([0-9A-F]+):',x); a=int(m.group(1),16)-base
                inner=[(t,c) for t,c in loops if t<=a<=c]
                sz=min([c-t for t,c in inner]) if inner else 0
                print("  +0x%05x %s  %s" % (a, x.strip().split('//')[0].strip()[:60], ("in loop of %d bytes" % sz) if inner else "outside loops"))
        print(" loops:",sorted(set(loops))[:12])
