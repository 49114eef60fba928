"""bf16 MFMA issue peak (v_mfma_f32_32x32x16_bf16, register-only loop) at 1..4 workgroups per CU, and for how long it holds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd._lib import call, ptr, stream_ptr
scratch = torch.empty(1024 * 256 * 4, device='cuda')
for blocks, it in ((256, 4000), (512, 4000), (768, 4000), (1024, 4000), (512, 40000), (512, 400000)):
    for _ in range(2):
        call('ssg_tool_mfma_peak_bf16', ptr(scratch), blocks, it, stream_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        call('ssg_tool_mfma_peak_bf16', ptr(scratch), blocks, it, stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print('%4d workgroups (%d per CU) x %d: %.2f ms  %.1f TFLOP/s' % (blocks, blocks // 256, it, ms, blocks * 4 * it * 16 * 32768 / ms / 1e9), flush=True)

data = torch.randn(4096 * 8, device='cuda').to(torch.bfloat16)
for blocks, it in ((512, 40000), (768, 40000), (512, 400000), (256, 400000)):
    for _ in range(2):
        call('ssg_tool_mfma_peak_bf16_data', ptr(scratch), blocks, it, ptr(data), stream_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        call('ssg_tool_mfma_peak_bf16_data', ptr(scratch), blocks, it, ptr(data), stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print('random operands, %4d workgroups (%d per CU) x %d: %.2f ms  %.1f TFLOP/s' % (blocks, blocks // 256, it, ms, blocks * 4 * it * 16 * 32768 / ms / 1e9), flush=True)
