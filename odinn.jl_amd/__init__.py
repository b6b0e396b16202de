"""odinn.jl_amd -- MI355X-native SIA2D(+NN_theta) time stepping and discrete adjoint,
the hot path of ODINN.jl, behind the C ABI of include/odinn_hip.h.

Import as ``odinn_jl_amd`` through ``_odinn_import.load()`` (the directory name is fixed
by the build contract and is not a Python identifier).
"""
from . import _lib
from ._lib import (ACT_GELU, ACT_IDENTITY, ACT_RELU, ACT_SIGMOID, ACT_SOFTPLUS, ACT_TANH, LAW_CONST_A,
                   LAW_NN_A_GRIDDED, LAW_NN_A_SCALAR, LAW_NN_U, LAW_NN_Y, POST_AFFINE, POST_EXPMAX, POST_NONE,
                   POST_SCALE, OdinnError, device_count, device_name)
from .batch import Comm, GlacierBatch, MLPSpec, PhysicalParameters, SolveStats
from .api import *  # noqa: F401,F403  (reference-facing names: Model, SIA2Dmodel, Prediction, Inversion, run_b ...)
