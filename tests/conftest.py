import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


@pytest.fixture(scope='session')
def golden():
  import numpy as np

  def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

  return load


@pytest.fixture(scope='session')
def gpu():
  """Loud failure when a gpu-marked test runs without the HIP path."""
  import torch
  from sofima_amd import _abi
  _abi.load()
  assert torch.cuda.is_available(), 'gpu tests need a visible MI355X'
  return torch.device('cuda', 0)
