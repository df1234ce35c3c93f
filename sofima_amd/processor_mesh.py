"""The relaxation driver of a section: regular / prepared / regularised passes.

Drop-in for the compute part of `processor/mesh.py: RelaxMesh.relax_mesh`
(processor/mesh.py:428-513), the step right after the hot path (SURVEY.md 8f,
rank 4): relax; if `map_utils.mask_irregular` finds folds, relax a fresh mesh
towards the first solution with k0 / 10, and if that one is regular, relax it
again against the real targets.  All three relaxations and both fold tests run
on the device; the mesh state stays in HBM between them and goes to the host
once, at the end.

The chunking / volume I/O of the `RelaxMesh` processor (connectomics
subvolumes, TensorStore) stays out of scope; it calls this function where it
called its method.
"""
from __future__ import annotations

import dataclasses
import enum
import logging

import numpy as np

from . import flow_utils
from . import map_utils
from . import mesh as mesh_lib


class SolutionStatus(enum.IntEnum):
  """processor/mesh.py:41-45."""
  UNDEFINED = -1
  REGULAR = 0
  PREP_FAILED = 1
  REGULARIZED = 2


class MeshInitState(enum.Enum):
  """processor/mesh.py:48-50."""
  ZEROS = 0
  PREV_MEDIAN = 1


def maybe_update_init_state(x: np.ndarray, prev: np.ndarray | None,
                            init_state: MeshInitState = MeshInitState.ZEROS
                            ) -> np.ndarray:
  """processor/mesh.py:387-398: optionally start from the median offset of the
  reference section."""
  if init_state == MeshInitState.PREV_MEDIAN and prev is not None:
    prev = np.asarray(prev)
    x[0, ...] = np.nanmedian(prev[0, ...])
    x[1, ...] = np.nanmedian(prev[1, ...])
    x = np.nan_to_num(x)
  return x


def relax_mesh(x: np.ndarray, prev: np.ndarray,
               integration_config: mesh_lib.IntegrationConfig,
               mask: np.ndarray | None = None, mesh_min_frac: float = 0.5,
               init_state: MeshInitState = MeshInitState.ZEROS
               ) -> tuple[np.ndarray, list[float], int, SolutionStatus]:
  """Performs mesh relaxation with the fold-recovery passes
  (processor/mesh.py:428-513).

  x, prev: [2, 1, y, x]; mask: optional [1, y, x] boolean, True entries of x
  are set to NaN (in place, like the reference).  Returns (optimised positions
  [np.ndarray], kinetic-energy history, steps simulated, SolutionStatus).

  `prev` may be a DeviceArray of an earlier call.  Like the reference,
  `mask_irregular` works IN PLACE on the solution it tests (map_utils.py:737-786
  modifies its argument, a view of x); PREP_FAILED returns the copy taken
  before the first test.
  """
  from . import _dev
  if mask is not None:
    flow_utils.apply_mask(x, mask)
  logging.info('Starting mesh relaxation.')
  dev = _dev.device()
  prev_d = _dev.as_device_f32(prev, dev, copy=False)      # uploaded once
  stride = integration_config.stride
  x_d, e_kin, num_steps = mesh_lib.relax_mesh(x, prev_d, integration_config)
  orig_x = x_d.tensor.clone()
  # the fold test NaNs the irregular nodes of the solution in place, on the device
  masked = map_utils.mask_irregular(x_d.tensor[:, 0], stride, mesh_min_frac,
                                    dilation_iters=5)
  if not np.any(masked):
    return np.array(x_d), e_kin, num_steps, SolutionStatus.REGULAR

  logging.info('Attempting relaxation with 10% k0.')
  # A new initial state that is similar to the previous solution everywhere
  # except near the irregular nodes; if it relaxes to a regular mesh, simulate
  # again from there, otherwise return the original solution.
  start_x = maybe_update_init_state(np.zeros(x_d.shape, np.float32), prev, init_state)
  x_d, _, prep_steps = mesh_lib.relax_mesh(
      start_x, x_d,
      dataclasses.replace(integration_config, k0=integration_config.k0 / 10.0))
  masked = map_utils.mask_irregular(x_d.tensor[:, 0], stride, mesh_min_frac)
  if np.any(masked):
    return (orig_x.cpu().numpy(), e_kin, num_steps + prep_steps,
            SolutionStatus.PREP_FAILED)

  if mask is not None:
    m = _dev.as_device_mask(mask, dev).bool()
    x_d.tensor[:, m] = float('nan')
  x_d, e_kin2, reg_steps = mesh_lib.relax_mesh(x_d, prev_d, integration_config)
  return (np.array(x_d), e_kin2, num_steps + prep_steps + reg_steps,
          SolutionStatus.REGULARIZED)
