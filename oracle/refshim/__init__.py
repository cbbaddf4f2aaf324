"""Import shim for the *reference* (TEST INFRASTRUCTURE -- this container only).

``install()`` makes ``smart_buildings.smart_control.*`` importable from
``/root/reference`` in an image that lacks gin / absl / cv2 / pint / holidays /
tensorflow / tf_agents / protoc.  It is used ONLY by ``oracle/gen_golden.py`` (to
produce the committed fixtures under ``tests/golden``) and by the optional
``tests/test_oracle_vs_reference.py`` (skipped when ``/root/reference`` is absent).
Nothing here is imported by the product (``sbsim_amd``) or travels to the GPU box
in any form that contains reference code: the shim only *points at* the reference
tree and supplies stand-ins for missing third-party modules.

Stand-ins and their fidelity:
  * gin.configurable            identity decorator (construction is done in Python).
  * absl.logging                -> stdlib logging.
  * holidays.US()               -> empty mapping (goldens avoid US holidays; the
                                   product implements federal holidays itself).
  * pint.UnitRegistry           unit objects of magnitude 1; quantities are bare
                                   numpy values (pint never rescales in the reference:
                                   every product is magnitude * magnitude).
  * cv2.connectedComponentsWithStats -> scipy.ndimage.label (4-connectivity, raster
                                   order); cv2.distanceTransform(DIST_L2, 3) -> 3x3
                                   chamfer (0.955, 1.3693) two-pass transform, the
                                   same mask OpenCV uses; cv2.dilate(iterations=0)
                                   -> identity.
  * *_pb2                        real protobuf classes built by ``protoc_lite`` from
                                   the reference's own .proto files (fp32 ``float``).
  * tf_simulator / building_renderer / visual_logger / mediapy / seaborn / tf_agents
                                   inert stand-ins (never on the numeric path).
"""
from __future__ import annotations

import importlib
import logging as _pylogging
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("SBSIM_REFERENCE_ROOT", "/root/reference")
_INSTALLED = False


def available() -> bool:
  return os.path.isdir(os.path.join(REFERENCE_ROOT, "smart_control", "simulator"))


def _module(name: str, **attrs) -> types.ModuleType:
  mod = types.ModuleType(name)
  mod.__dict__.update(attrs)
  sys.modules[name] = mod
  return mod


# ----------------------------------------------------------------------------- gin
def _configurable(*args, **kwargs):
  if len(args) == 1 and callable(args[0]) and not kwargs:
    return args[0]
  return lambda fn: fn


# ----------------------------------------------------------------------------- pint
class _Quantity:
  """Magnitude-only stand-in for pint.Quantity (units are all magnitude 1)."""
  __array_priority__ = 1000
  __array_ufunc__ = None

  def __init__(self, value):
    self.value = value

  @property
  def magnitude(self):
    return self.value

  @staticmethod
  def _v(other):
    return other.value if isinstance(other, _Quantity) else other

  def __mul__(self, other):
    return _Quantity(self.value * self._v(other))

  def __rmul__(self, other):
    other = self._v(other)
    if isinstance(other, (tuple, list)):
      other = np.array(other)
    return _Quantity(other * self.value)

  def __truediv__(self, other):
    return _Quantity(self.value / self._v(other))

  def __getitem__(self, idx):
    return _Quantity(self.value[idx])

  def __len__(self):
    return len(self.value)

  def __array__(self, dtype=None, copy=None):
    return np.asarray(self.value, dtype=dtype)

  def __iter__(self):
    return iter(self.value)


class _UnitRegistry:

  def define(self, _spec: str) -> None:
    return None

  def __getattr__(self, _name: str) -> _Quantity:
    return _Quantity(1.0)


# ----------------------------------------------------------------------------- cv2
def _connected_components_with_stats(img, connectivity=4):
  from scipy import ndimage
  if connectivity == 4:
    structure = ndimage.generate_binary_structure(2, 1)
  else:
    structure = np.ones((3, 3), dtype=bool)
  labels, n = ndimage.label(np.asarray(img) != 0, structure=structure)
  return n + 1, labels.astype(np.int32), None, None


def _distance_transform(src, _dist_type, _mask_size):
  """3x3 chamfer transform with OpenCV's DIST_L2 weights (0.955, 1.3693)."""
  a, b = np.float32(0.955), np.float32(1.3693)
  src = np.asarray(src)
  h, w = src.shape
  big = np.float32(1e9)
  d = np.where(src == 0, np.float32(0), big).astype(np.float32)
  pad = np.full((h + 2, w + 2), big, dtype=np.float32)
  pad[1:-1, 1:-1] = d
  for i in range(1, h + 1):
    for j in range(1, w + 1):
      if pad[i, j] == 0:
        continue
      pad[i, j] = min(pad[i, j], pad[i - 1, j - 1] + b, pad[i - 1, j] + a,
                      pad[i - 1, j + 1] + b, pad[i, j - 1] + a)
  for i in range(h, 0, -1):
    for j in range(w, 0, -1):
      if pad[i, j] == 0:
        continue
      pad[i, j] = min(pad[i, j], pad[i + 1, j + 1] + b, pad[i + 1, j] + a,
                      pad[i + 1, j - 1] + b, pad[i, j + 1] + a)
  return pad[1:-1, 1:-1]


def _dilate(src, _kernel, iterations=1):
  if iterations != 0:
    raise NotImplementedError("refshim cv2.dilate only supports iterations=0")
  return src


# ----------------------------------------------------------------------------- inert
class _Inert:

  def __init__(self, *a, **k):
    pass

  def __getattr__(self, _name):
    return lambda *a, **k: None


def install() -> None:
  """Installs the stand-ins and the ``smart_buildings`` alias (idempotent)."""
  global _INSTALLED
  if _INSTALLED:
    return
  if not available():
    raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")

  _module("gin", configurable=_configurable, REQUIRED=object(),
          external_configurable=lambda fn, *a, **k: fn)

  log = _pylogging.getLogger("refshim.absl")
  log.setLevel(_pylogging.ERROR)
  absl_logging = _module(
      "absl.logging", info=log.info, warning=log.warning, warn=log.warning,
      error=log.error, exception=log.exception, debug=log.debug, fatal=log.critical)
  # absl.testing: just enough to import the reference's *_test.py modules and call
  # their fixture helpers (_create_scenario_building, _create_scenario_hvac, ...).
  import unittest

  def _params_decorator(*_a, **_k):
    return lambda fn: fn

  absltest = _module("absl.testing.absltest", TestCase=unittest.TestCase,
                     main=unittest.main)
  parameterized = _module(
      "absl.testing.parameterized", TestCase=unittest.TestCase,
      named_parameters=_params_decorator, parameters=_params_decorator,
      product=_params_decorator)
  absl_testing = _module("absl.testing", absltest=absltest, parameterized=parameterized)
  _module("absl", logging=absl_logging, testing=absl_testing)

  _module("holidays", US=lambda *a, **k: {})
  _module("pint", UnitRegistry=_UnitRegistry)
  _module("cv2", connectedComponentsWithStats=_connected_components_with_stats,
          distanceTransform=_distance_transform, dilate=_dilate, DIST_L2=2)
  _module("mediapy")
  _module("seaborn")
  # utils/writer_lib.py only names the type in an annotation
  ir_abc = _module("importlib_resources.abc", Traversable=object)
  _module("importlib_resources", abc=ir_abc)

  # tf_agents.specs is only touched to build ArraySpecs; give it inert constructors.
  specs = _module("tf_agents.specs", BoundedArraySpec=_Inert, ArraySpec=_Inert)
  _module("tf_agents", specs=specs)

  # google3.google.protobuf.timestamp_pb2 -> the public one.
  from google.protobuf import timestamp_pb2
  g3p = _module("google3.google.protobuf", timestamp_pb2=timestamp_pb2)
  g3g = _module("google3.google", protobuf=g3p)
  _module("google3", google=g3g)
  sys.modules["google3.google.protobuf.timestamp_pb2"] = timestamp_pb2

  # smart_buildings -> /root/reference
  pkg = _module("smart_buildings")
  pkg.__path__ = [REFERENCE_ROOT]
  sc = _module("smart_buildings.smart_control")
  sc.__path__ = [os.path.join(REFERENCE_ROOT, "smart_control")]
  for sub in ("simulator", "reward", "models", "utils", "environment"):
    m = _module(f"smart_buildings.smart_control.{sub}")
    m.__path__ = [os.path.join(REFERENCE_ROOT, "smart_control", sub)]
    setattr(sc, sub, m)

  # real protobuf classes from the reference's own .proto files
  from . import protoc_lite
  proto_pkg = _module("smart_buildings.smart_control.proto")
  proto_pkg.__path__ = []
  proto_dir = os.path.join(REFERENCE_ROOT, "smart_control", "proto")
  for stem in ("smart_control_building", "smart_control_reward",
               "smart_control_normalization"):
    name = f"smart_buildings.smart_control.proto.{stem}_pb2"
    mod = protoc_lite.build_module(
        os.path.join(proto_dir, stem + ".proto"), name,
        f"refshim/{stem}.proto")
    sys.modules[name] = mod
    setattr(proto_pkg, stem + "_pb2", mod)

  # modules that drag in tensorflow / plotting: inert stand-ins
  base = "smart_buildings.smart_control"
  _module(f"{base}.simulator.tf_simulator", TFSimulator=_Inert)
  _module(f"{base}.utils.building_renderer", BuildingRenderer=_Inert)
  _module(f"{base}.utils.visual_logger", VisualLogger=_Inert)
  _INSTALLED = True


def ref(module: str):
  """Imports ``smart_buildings.smart_control.<module>`` from the reference."""
  install()
  return importlib.import_module(f"smart_buildings.smart_control.{module}")
