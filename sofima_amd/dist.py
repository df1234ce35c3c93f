"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed.

The path shards by INDEPENDENT units and needs no data-path collective
(SURVEY.md 8e): the patches of one flow field are split by whole batches
(batch membership must not change, because the reference's second-peak
suppression and masked-NCC tolerances are batch-coupled), tile pairs and
section pairs are split as units, and every rank relaxes its own meshes.  The
only communication is the gather of the (small) results.

Backends: "nccl" (= RCCL over xGMI) on GPUs; "gloo" for the CPU tests, which
replace the device call by a host function (`batch_fn`) to check that sharded
and single-process results are identical.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import torch.distributed as dist


def world(group=None) -> tuple[int, int]:
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(group), dist.get_world_size(group)
  return 0, 1


def shard_units(n_units: int, rank: int, world_size: int) -> list[int]:
  """Round-robin assignment of unit indices (batches, tile pairs, sections)."""
  return list(range(rank, n_units, world_size))


def gather_objects(obj, group=None) -> list:
  rank, ws = world(group)
  if ws == 1:
    return [obj]
  out = [None] * ws
  dist.all_gather_object(out, obj, group=group)
  return out


def sharded_flow_field(calc, pre_image, post_image, patch_size, step, *,
                       pre_mask=None, post_mask=None,
                       mask_only_for_patch_selection=False,
                       selection_mask=None, max_masked=0.75, batch_size=4096,
                       post_patch_size=None, pre_targeting_field=None,
                       pre_targeting_step=None, post_targeting_field=None,
                       post_targeting_step=None, group=None,
                       batch_fn: Callable | None = None) -> np.ndarray:
  """`calc.flow_field(...)` with the batches of patches spread over the ranks.

  Every rank must hold the same images (or at least the rows its batches
  touch) and passes the same arguments; every rank returns the full field,
  bit-identical to the single-process result.

  batch_fn(pre_starts, post_starts) -> peaks [len, dim + 2] replaces the GPU
  call (CPU tests).
  """
  nd = pre_image.ndim
  seq = lambda v: tuple(int(a) for a in v) if np.ndim(v) else (int(v),) * nd
  patch_size = seq(patch_size)
  post_patch_size = patch_size if post_patch_size is None else seq(
      post_patch_size)
  step = seq(step)
  if pre_targeting_step is not None:
    pre_targeting_step = seq(pre_targeting_step)
  if post_targeting_step is not None:
    post_targeting_step = seq(post_targeting_step)
  plan = calc.plan(tuple(pre_image.shape), tuple(post_image.shape), patch_size,
                   step, pre_mask, post_mask, selection_mask, max_masked,
                   batch_size, post_patch_size, pre_targeting_field,
                   pre_targeting_step, post_targeting_field,
                   post_targeting_step)
  rank, ws = world(group)
  mine = shard_units(plan['n_batches'], rank, ws)
  if mask_only_for_patch_selection:
    pre_mask = post_mask = None
  if batch_fn is None:
    local = calc.compute_batches(pre_image, post_image, pre_mask, post_mask,
                                 patch_size, post_patch_size, plan, batch_size,
                                 mine)
  else:
    parts = [
        batch_fn(plan['pre_starts'][b * batch_size:(b + 1) * batch_size],
                 plan['post_starts'][b * batch_size:(b + 1) * batch_size])
        for b in mine
    ]
    local = (np.concatenate(parts) if parts else
             np.zeros((0, nd + 2), np.float32))
  peaks = np.zeros((plan['n_batches'] * batch_size, nd + 2), np.float32)
  for r, part in enumerate(gather_objects(np.asarray(local), group)):
    for k, b in enumerate(shard_units(plan['n_batches'], r, ws)):
      peaks[b * batch_size:(b + 1) * batch_size] = part[k * batch_size:
                                                        (k + 1) * batch_size]
  return calc.assemble(plan, nd, peaks)


def map_units(units: Sequence, fn: Callable, group=None) -> list:
  """Applies `fn` to this rank's share of independent `units` (section pairs,
  tile pairs, meshes) and returns the results of ALL units, in order, on
  every rank."""
  rank, ws = world(group)
  mine = shard_units(len(units), rank, ws)
  local = [fn(units[i]) for i in mine]
  out = [None] * len(units)
  for r, part in enumerate(gather_objects(local, group)):
    for k, i in enumerate(shard_units(len(units), r, ws)):
      out[i] = part[k]
  return out
