import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
from okvis2_amd import capi, synth
for name,cfgf in (("euroc",synth.euroc_config),("mono640",synth.mono640_config)):
    cfg=cfgf(); C=len(cfg.cams); B=32
    imgs,_=bench.make_inputs(cfg,B,16,1000,"corners")
    fe=capi.Frontend(cfg.w,cfg.h,cfg.uniformity_radius,cfg.octaves,cfg.abs_threshold,cfg.max_kpts,max_batch=C*B,num_cameras=C)
    for ci,cam in enumerate(cfg.cams): fe.set_camera(ci,cam)
    d=torch.from_numpy(imgs).cuda()
    cam_ids=np.array(list(range(C))*B,dtype=np.int32)
    fe.detect_describe_batch_device(d.data_ptr(),C*B,cam_ids,bench.gravity_variant(0,C*B),torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out=fe.device_outputs()
    import ctypes
    def rd(ptr,n):
        a=np.empty(n,dtype=np.int32); 
        t=torch.empty(n,dtype=torch.int32,device='cuda')
        ctypes.cdll.LoadLibrary('libamdhip64.so').hipMemcpy(ctypes.c_void_p(a.ctypes.data),ctypes.c_void_p(ptr),ctypes.c_size_t(4*n),ctypes.c_int(2))
        return a
    print(name,"detect",rd(out.detect_counts,C*B).mean(),"valid",rd(out.counts,C*B).mean(),"cand",rd(out.candidate_counts,C*B).mean())
