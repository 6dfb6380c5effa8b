import sys; sys.path.insert(0,'.')
import numpy as np, ctypes as C
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
from oracle import pyoracle as O
d=np.load('tests/golden/tie_blocks.npz')
img=d['img']
ctx=T.Context(0)
zz=T.dctq(img,50,ctx=ctx)
want=O.encode_zz16(img,50)
bad=np.argwhere(zz!=want)
print('mismatches',len(bad))
for b,k in bad[:20]:
    print('block',b,'zz',k,'got',zz[b,k],'want',want[b,k])
