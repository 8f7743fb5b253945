import sys, os
sys.argv=[sys.argv[0],'16','attn']
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'tests'), os.path.join(ROOT,'tools')]
import builtins
# reuse x3_probe's attention() by exec'ing its source up to the first call
src=open(os.path.join(ROOT,'tools','x3_probe.py')).read()
src=src[:src.index('T = 1500\nattention(')]
exec(compile(src,'x3_probe','exec'))
T=1500
attention(256,1,T,"256 seq x 1 head")
attention(16,16,T,"16 seq x 16 heads")
attention(512,1,T,"512 seq x 1 head")
attention(32,16,T,"32 seq x 16 heads")
attention(64,8,T,"64 seq x 8 heads")
