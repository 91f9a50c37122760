import sys; sys.path.insert(0,'.')
import numpy as np, ga3c_amd, _native as nat
from NetworkVP import Network
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
NL=int(sys.argv[2]) if len(sys.argv)>2 else 4
net=Network("gpu:0","l",6,(84,84,4),max_batch=B,predict_lanes=NL)
x=np.random.default_rng(0).integers(0,256,size=(B,84,84,4),dtype=np.uint8).astype(np.float32)/128-1
xk=np.random.default_rng(0).integers(0,256,size=(B,84,84,4),dtype=np.uint8)
nat.check(net._lib.ga3c_net_upload_u8(net._h,nat.ptr(xk,nat.u8p),None,None,B))
ms=nat.C.c_float()
for nl in range(1,NL+1):
    nat.check(net._lib.ga3c_net_time_predict_lanes(net._h,B,50,nl,nat.C.byref(ms)))
    nat.check(net._lib.ga3c_net_time_predict_lanes(net._h,B,400,nl,nat.C.byref(ms)))
    print("lanes %d: %.2f us per step -> %.2f M pred/s"%(nl, ms.value/400*1e3, 400*B/ms.value/1e3))
net.close()
