"""TEST INFRASTRUCTURE: run the product's Python host code on CPU tensors with libmedt_emu.so (the whole kernel library compiled
for the CPU lane emulator, tests/lane_emu) standing in for libmedt_hip.so -- an emulated device for tests, NOT a CPU path of the
product: outside this context manager the product refuses CPU tensors as always (tests/test_abi.py)."""
import contextlib
import types

import torch


class DeviceTensor(torch.Tensor):
    """A CPU tensor that answers is_cuda like the device tensors the product insists on."""
    is_cuda = property(lambda self: True)


@contextlib.contextmanager
def emulated_device(emu):
    from medt_amd import _lib as L, axial, block, net, ops
    stream = types.SimpleNamespace(cuda_stream=0, wait_stream=lambda s: None, wait_event=lambda e: None)
    saved = []

    def patch(obj, name, value):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    fused = block.fused_forward
    try:
        patch(L, "_lib", emu)                                     # L.lib() hands out the emulated library
        for mod in (axial, ops, net):
            patch(mod, "_require_device", lambda x: None)
        patch(torch.cuda, "current_stream", lambda *a, **k: stream)
        patch(torch.cuda, "is_current_stream_capturing", lambda: False)
        patch(torch.cuda, "synchronize", lambda *a, **k: None)
        patch(net, "TWO_STREAMS", False)                          # (streams are a scheduling matter; one queue, same arithmetic)
        patch(block, "fused_forward", lambda blk, x, g: fused(blk, x.as_subclass(DeviceTensor), g))
        from medt_amd import optim
        launch_adam = optim.FlatAdam._launch_adam

        def launch_adam_emulated(self, g, gscale):              # FlatAdam's own device check: its flat buffers as device tensors
            keep = g.flat_p
            g.flat_p = keep.as_subclass(DeviceTensor)
            try:
                return launch_adam(self, g, gscale)
            finally:
                g.flat_p = keep
        patch(optim.FlatAdam, "_launch_adam", launch_adam_emulated)
        yield
    finally:
        for obj, name, value in reversed(saved):
            setattr(obj, name, value)
