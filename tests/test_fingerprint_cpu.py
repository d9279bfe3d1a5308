"""FramePipeline._fingerprint: the change detector behind model.pipeline() (called several times per rendered frame) reads the parameters and buffers from a cached
module list instead of walking the module tree through nn.Module.parameters(); it must still see everything the walk saw."""
import torch
import torch.nn as nn

from genefaceplusplus_amd.radnerfs.frame_pipeline import FramePipeline


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Sequential(nn.Linear(4, 8), nn.ReLU(), nn.Linear(8, 2))
        self.b = nn.Linear(2, 2, bias=False)
        self.register_buffer("table", torch.zeros(5))
        self.register_buffer("nothing", None)


def test_fingerprint_sees_what_the_module_walk_saw():
    net = _Net()
    fp = FramePipeline._fingerprint
    walked = tuple((t.data_ptr(), t._version) for t in list(net.parameters()) + list(net.buffers()))
    assert sorted(fp(net)) == sorted(walked)                     # the same tensors (order aside)
    base = fp(net)
    assert fp(net) == base                                       # stable
    with torch.no_grad():
        net.a[2].weight.add_(1.0)                                # an optimizer step / load_state_dict: in place
    assert fp(net) != base
    base = fp(net)
    net.table.zero_()                                            # a buffer written in place
    assert fp(net) != base
    base = fp(net)
    net.b.weight = nn.Parameter(torch.ones(2, 2))                # a replaced Parameter object
    assert fp(net) != base
    base = fp(net)
    net.a[0] = nn.Linear(4, 8)                                   # a replaced sub-module (same slot, same count of children)
    assert fp(net) != base
    base = fp(net)
    net.c = nn.Linear(1, 1)                                      # an added sub-module
    assert len(fp(net)) == len(base) + 2
    base = fp(net)
    net.b.weight.data = torch.zeros(2, 2)                        # .to() / .half(): the parameter's storage moves
    assert fp(net) != base
    base = fp(net)
    net.register_buffer("late", torch.ones(1))                   # an added buffer on an existing module
    assert len(fp(net)) == len(base) + 1
