"""Launched by tests/test_gpu_rccl.py under torch.distributed.run on ONE GPU: initialises the RCCL ('nccl') backend the
way bench.py does, runs the packed variable-length all-gather on the one-rank communicator and the slide loop through the
distributed code path, and compares with the non-distributed result."""
import os
import sys

import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from celldetection_amd import inference
    from test_gpu_model import build
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    td.init_process_group('nccl', device_id=dev)
    assert td.get_world_size() == 1 and td.get_backend() == 'nccl'
    buf = torch.rand(37, 161, device=dev)
    out = inference.gather_detections(buf, _force=True)  # counts all-gather + padded payload all-gather on RCCL
    assert torch.equal(out, buf)
    empty = inference.gather_detections(buf[:0], _force=True)
    assert empty.shape[0] == 0
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MAX)  # the reduction bench.py uses for the step time
    td.barrier()
    model, g = build('CpnU22', dev, fixture='stitch.npz')
    img = torch.as_tensor(g['img']).to(dev)
    a = inference.tiled_inference(model, img, (96, 96), (64, 64), batch_size=4)  # world/rank from torch.distributed
    b = inference.tiled_inference(model, img, (96, 96), (64, 64), batch_size=4, rank=0, world_size=1)
    for k in inference.KEYS:
        assert torch.equal(a[k], b[k]), k
    td.destroy_process_group()
    print('RCCL_PROBE_OK', int(a['scores'].numel()))


if __name__ == '__main__':
    main()
