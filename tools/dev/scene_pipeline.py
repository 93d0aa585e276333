"""EXPERIMENTAL -- independent reference views on several HIP streams of one GPU.

Why: a scene is ~100 dependent launches; about a third of its time goes to the deep U-Net levels and small glue kernels -- launches
of 10-30 us that occupy a fraction of the 256 CUs -- while the big layers fill the chip.  Two scenes in flight on two streams let one
scene's small launches run beside the other's large ones: 1.38 -> 1.14-1.28 ms per scene eager (host-bound, it varies with the box;
`bench.py --streams 2`), 1.08 ms with hipGraph replay (bench.py --in-flight-side-pass),
each scene still one ``CascadeMVSNet_eval.forward`` at batch 1.  One model replica per stream: a model's activation-bound buffer
(fp16-pair form) and plan caches are not re-entrant.

STATUS: OFF, and it stays off.  Round 3 (profiles/r3_two_streams.txt): at the full config-2 size this mode corrupted the STAGE-3 outputs of
~7 % of the scenes (tile-shaped patches); the first wrong tensor of every corrupted scene was stage 3's hypothesis planes, which had read
32-byte pieces of the PREVIOUS contents of the reused allocator block holding stage 2's depth map although the writer had completed on
the same stream; agent-scope (sc1) loads in that one reader removed it (0 corrupted in ~7000 scenes; adopted, csrc/geometry.hip).
Round 4 (profiles/r4_two_streams_ab.txt): an explicit agent-scope acquire at the top of EVERY inference kernel, with that reader back on
plain loads, does NOT help (84 of 90 rounds corrupted) and costs 18 % on one stream -- the stale data is not in the reader's vector L1;
the kernel-boundary coherence between a writer and a reader of one queue is not reliable while a second queue of the same process is
dispatching, and nothing guarantees that the planes kernel is the only exposed reader.  Hence: one stream is the supported mode;
``ScenePipeline(nstreams > 1)`` refuses to start unless the caller passes ``experimental=True`` (or RCMVS_ALLOW_STREAMS=1), and
``bench.py --streams N`` self-checks its outputs against the one-stream run and says so in the line.  Two worker PROCESSES per GPU give
the same gain (912 vs 731 ref-scenes/s, bench.py's `two_procs_per_gpu`; rc_mvsnet_amd/sharding.py, eval_driver --procs-per-gpu 2)
without sharing a runtime."""

import os

import torch


class ScenePipeline:
    def __init__(self, make_model, nstreams=1, device=None, wait_inputs=True, experimental=False):
        """make_model() -> an eval-mode module on `device`; called `nstreams` times.  wait_inputs=False: the caller guarantees that the
        inputs are complete before the call (resident, synchronised), which saves an event per scene."""
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = max(1, int(nstreams))
        if self.n > 1 and not (experimental or os.environ.get("RCMVS_ALLOW_STREAMS") == "1"):
            raise RuntimeError("ScenePipeline: more than one HIP stream per process is an experimental mode with a known, unexplained data hazard "
                               "(see the module docstring); pass experimental=True (or RCMVS_ALLOW_STREAMS=1) and check the outputs against a "
                               "one-stream run -- or use two worker processes per GPU (rc_mvsnet_amd.sharding, --procs-per-gpu 2)")
        self.models = [make_model() for _ in range(self.n)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)] if self.n > 1 else [None]
        self._i = 0
        self.wait_inputs = bool(wait_inputs)
        if self.n > 1:
            import warnings
            warnings.warn("ScenePipeline with more than one stream is experimental: the stage-3 corruption is fixed at its first wrong op only, "
                          "its cause is not understood (see the module docstring); check the outputs against a one-stream run",
                          RuntimeWarning, stacklevel=2)

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
