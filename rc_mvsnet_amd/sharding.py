"""Multi-GPU inference = independent (scene, reference-view) items, one process per GPU, no
data-path collective: each rank takes a round-robin slice of the item list
(what eval_rcmvsnet_dtu.py:157-165 iterates serially on one GPU)."""


def shard_items(items, rank, world_size):
    """Round-robin partition: rank r gets items r, r+W, r+2W, ...  (balanced to within one item)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    return list(items[rank::world_size])
