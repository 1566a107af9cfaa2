"""Collectives of the view-parallel paths behind one switch: RCCL on the device, anything else on the host."""
import torch


def all_reduce_(t: torch.Tensor, op=None, group=None) -> None:
    """In-place all-reduce of a device tensor.  RCCL ("nccl") reduces on the device; any other transport
    (gloo, which the single-GPU tests use) is given HOST tensors: its own device path stages through pinned
    memory on side streams and has stalled when two ranks share one GPU, a copy to the host and back has not."""
    op = torch.distributed.ReduceOp.SUM if op is None else op
    if t.device.type != "cuda" or torch.distributed.get_backend(group) == "nccl":
        torch.distributed.all_reduce(t, op=op, group=group)
        return
    h = t.detach().cpu()
    torch.distributed.all_reduce(h, op=op, group=group)
    t.copy_(h)
