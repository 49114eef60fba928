"""fp32 MFMA issue peak at 1, 2, 3, 4 workgroups (waves per SIMD) per CU: register-only loop, no LDS, no barriers."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd._lib import call, ptr, stream_ptr
scratch = torch.empty(1024 * 256 * 4, device='cuda')
it = 20000
for blocks in (256, 512, 768, 1024, 256, 768):
    for _ in range(2):
        call('ssg_tool_mfma_peak_f32', ptr(scratch), blocks, it, stream_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        call('ssg_tool_mfma_peak_f32', ptr(scratch), blocks, it, stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print('%4d workgroups (%d per CU): %.2f ms  %.1f TFLOP/s' % (blocks, blocks // 256, ms, blocks * 4 * it * 16 * 4096 / ms / 1e9), flush=True)
