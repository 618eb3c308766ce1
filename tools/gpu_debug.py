import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import helpers as H
from tokendagger_amd import capi
g=np.load('tests/golden/llama4_golden.npz',allow_pickle=True)
pat,mr,sp=H.llama4()
tok=capi.HipTokenizer(pat,mr,sp,device=0)
text,offs=g['text'],g['offsets']; enc,eo=g['enc'],g['enc_offsets']; names=g['names']
toks,toffs=tok.encode_batch(text,offs)
print('batch total',len(toks),'expected',len(enc),'long pieces',tok.info(7))
nbad=0
for d in range(len(offs)-1):
    a=toks[toffs[d]:toffs[d+1]]; b=enc[eo[d]:eo[d+1]]
    if not np.array_equal(a,b):
        nbad+=1
        if nbad<=12:
            doc=text[offs[d]:offs[d+1]].tobytes()
            n=min(len(a),len(b)); k=int(np.argmax(a[:n]!=b[:n])) if n and (a[:n]!=b[:n]).any() else n
            print('BATCH MISMATCH',names[d],'len',len(doc),'got',len(a),'exp',len(b),'first diff',k,a[max(0,k-2):k+4],b[max(0,k-2):k+4],repr(doc[:60]))
print('batch mismatching docs',nbad)
nbad=0
for d in range(len(offs)-1):
    doc=text[offs[d]:offs[d+1]].tobytes()
    try:
        a=tok.encode(doc)
    except Exception as e:
        print('ERR',names[d],e); nbad+=1; continue
    b=enc[eo[d]:eo[d+1]]
    if not np.array_equal(a,b):
        nbad+=1
        if nbad<=12:
            n=min(len(a),len(b)); k=int(np.argmax(a[:n]!=b[:n])) if n and (a[:n]!=b[:n]).any() else n
            print('SINGLE MISMATCH',names[d],'len',len(doc),'got',len(a),'exp',len(b),'first diff',k,a[max(0,k-2):k+4],b[max(0,k-2):k+4],repr(doc[:60]))
print('single mismatching docs',nbad)
