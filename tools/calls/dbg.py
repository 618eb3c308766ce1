import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import helpers as H, td_corpus
from tokendagger_amd import capi
pat, mr, special = H.llama4()
t = capi.HipTokenizer(pat, mr, special, device=0)
t.set_option(capi.TD_OPT_SMALL_PATH, 0)
x, o = td_corpus.code_files(8 << 20)
try:
    a = t.encode_batch(x.tobytes(), o); print("code_files ok", len(a[0]))
except Exception as e:
    print("code_files FAILED", e)
