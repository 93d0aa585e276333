"""EXPERIMENTAL -- independent reference views on several HIP streams of one GPU.  NOT SAFE FOR RESULTS YET.

Why: a scene is ~100 dependent launches; about a third of its time goes to the deep U-Net levels and small glue kernels -- launches
of 10-30 us that occupy a fraction of the 256 CUs -- while the big layers fill the chip.  Two scenes in flight on two streams let one
scene's small launches run beside the other's large ones: 1.38 -> 1.14-1.21 ms per scene eager (`bench.py --streams 2`), 1.08 ms
with hipGraph replay (tools/dev/two_stream_graph_probe.py), each scene still one ``CascadeMVSNet_eval.forward`` at batch 1.
One model replica per stream: a model's activation-bound buffer (fp16-pair form) and plan caches are not re-entrant.

OPEN DEFECT (round 3, profiles/r3_two_streams.txt): at the full config-2 size the STAGE-3 outputs of some scenes come out wrong in
6-8 of 10 runs -- tile-shaped patches (64 x 32 pixels + 8 of spread, 1-25 % of the pixels), stages 1 and 2 always bit-identical, no
NaN.  It needs library kernels on BOTH streams (scenes next to foreign torch kernels on the other stream: 0 of 24 runs), any
device-wide synchronisation inside stage 3 hides it, AMD_SERIALIZE_KERNEL=3 hides it, and it is independent of the arithmetic form,
of the split-operand kernel of stage 3's conv0, of K1's variant, of out-of-range padding loads and of the LDS contents at kernel
start (all switched off one at a time, tools/dev/two_stream_*.py).  The replicas share no device memory that has been found.  Until
it is root-caused the one-stream loop is the only supported mode; ``bench.py --streams N`` self-checks its outputs and says so."""

import torch


class ScenePipeline:
    def __init__(self, make_model, nstreams=2, device=None, wait_inputs=True):
        """make_model() -> an eval-mode module on `device`; called `nstreams` times.  wait_inputs=False: the caller guarantees that the
        inputs are complete before the call (resident, synchronised), which saves an event per scene."""
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = max(1, int(nstreams))
        self.models = [make_model() for _ in range(self.n)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)] if self.n > 1 else [None]
        self._i = 0
        self.wait_inputs = bool(wait_inputs)
        if self.n > 1:
            import warnings
            warnings.warn("ScenePipeline with more than one stream is experimental: intermittently corrupted stage-3 outputs at full size "
                          "(see the module docstring); use one stream for results", RuntimeWarning, stacklevel=2)

    def __call__(self, *args, **kwargs):
        """Issue one scene on the next stream; returns (outputs, stream) -- the outputs are valid once `stream` (None = the current
        stream) has been waited for (``stream.synchronize()``, ``torch.cuda.current_stream().wait_stream(stream)``, or synchronize())."""
        k = self._i % self.n
        self._i += 1
        if self.streams[k] is None:
            return self.models[k](*args, **kwargs), None
        if self.wait_inputs:
            self.streams[k].wait_stream(torch.cuda.current_stream(self.device))      # inputs produced on the caller's stream are ready
        with torch.cuda.stream(self.streams[k]):
            return self.models[k](*args, **kwargs), self.streams[k]

    def synchronize(self):
        for s in self.streams:
            if s is not None:
                s.synchronize()
