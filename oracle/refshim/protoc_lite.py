"""Minimal proto3 -> message-class builder (TEST INFRASTRUCTURE, this container only).

The image has the protobuf *runtime* but no ``protoc`` and the reference does not
vendor its generated ``*_pb2`` modules.  This module parses a ``.proto`` file at run
time (straight from ``/root/reference/smart_control/proto/*.proto`` -- nothing is
copied into this repo) into a ``FileDescriptorProto`` and asks the protobuf runtime for
real message classes, so that the reference's ``float`` fields get genuine fp32
truncation and ``CopyFrom`` / ``WhichOneof`` / map fields behave exactly as they do
for the reference's own generated code.

Supported subset: proto3 messages (nested), enums (nested), scalar fields,
``repeated``, ``map<k, v>``, ``oneof``, message-typed fields, the
``google.protobuf.Timestamp`` import.  That is everything the three sbsim protos use.
"""
from __future__ import annotations

import re
import types
from typing import List

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
from google.protobuf import timestamp_pb2  # noqa: F401  (registers the dependency)

_F = descriptor_pb2.FieldDescriptorProto
_SCALARS = {
    "double": _F.TYPE_DOUBLE, "float": _F.TYPE_FLOAT, "int32": _F.TYPE_INT32,
    "int64": _F.TYPE_INT64, "uint32": _F.TYPE_UINT32, "uint64": _F.TYPE_UINT64,
    "sint32": _F.TYPE_SINT32, "sint64": _F.TYPE_SINT64, "bool": _F.TYPE_BOOL,
    "string": _F.TYPE_STRING, "bytes": _F.TYPE_BYTES,
}


def _tokenize(text: str) -> List[str]:
  text = re.sub(r"//[^\n]*", "", text)
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return re.findall(r'"[^"]*"|[A-Za-z_][A-Za-z0-9_.]*|-?\d+|[{}=;<>,\[\]()]', text)


class _Parser:

  def __init__(self, tokens: List[str], package_hint: str = ""):
    self.t = tokens
    self.i = 0
    self.package = package_hint

  def peek(self) -> str:
    return self.t[self.i]

  def next(self) -> str:
    tok = self.t[self.i]
    self.i += 1
    return tok

  def expect(self, tok: str) -> None:
    got = self.next()
    if got != tok:
      raise ValueError(f"proto parse: expected {tok!r}, got {got!r} at {self.i}")

  def skip_options(self) -> None:
    if self.peek() == "[":
      while self.next() != "]":
        pass

  # -- type resolution ----------------------------------------------------
  def _set_type(self, field, type_name: str, scope: List[str], known: dict):
    if type_name in _SCALARS:
      field.type = _SCALARS[type_name]
      return
    if type_name == "google.protobuf.Timestamp":
      field.type = _F.TYPE_MESSAGE
      field.type_name = ".google.protobuf.Timestamp"
      return
    # innermost-scope-first lookup of message / enum names
    for depth in range(len(scope), -1, -1):
      cand = ".".join(scope[:depth] + [type_name])
      if cand in known:
        field.type = _F.TYPE_ENUM if known[cand] == "enum" else _F.TYPE_MESSAGE
        field.type_name = "." + self.package + "." + cand
        return
    raise ValueError(f"proto parse: unknown type {type_name} in {scope}")

  # -- pre-pass: collect declared names ----------------------------------
  def collect(self) -> dict:
    known, scope, depth_stack = {}, [], []
    i = 0
    while i < len(self.t):
      tok = self.t[i]
      if tok in ("message", "enum") and self.t[i + 2] == "{":
        known[".".join(scope + [self.t[i + 1]])] = tok
        scope.append(self.t[i + 1])
        depth_stack.append("named")
        i += 3
        continue
      if tok == "{":
        depth_stack.append("anon")
      elif tok == "}":
        if depth_stack.pop() == "named":
          scope.pop()
      i += 1
    return known

  # -- grammar ------------------------------------------------------------
  def parse_enum(self, enum_proto) -> None:
    enum_proto.name = self.next()
    self.expect("{")
    while self.peek() != "}":
      name = self.next()
      self.expect("=")
      number = int(self.next())
      self.skip_options()
      self.expect(";")
      v = enum_proto.value.add()
      v.name, v.number = name, number
    self.expect("}")

  def parse_message(self, msg, scope: List[str], known: dict) -> None:
    msg.name = self.next()
    scope = scope + [msg.name]
    self.expect("{")
    while self.peek() != "}":
      tok = self.next()
      if tok == "message":
        self.parse_message(msg.nested_type.add(), scope, known)
      elif tok == "enum":
        self.parse_enum(msg.enum_type.add())
      elif tok == "oneof":
        oneof = msg.oneof_decl.add()
        oneof.name = self.next()
        idx = len(msg.oneof_decl) - 1
        self.expect("{")
        while self.peek() != "}":
          f = self.parse_field(msg, self.next(), scope, known)
          f.oneof_index = idx
        self.expect("}")
      elif tok in ("reserved", "option"):
        while self.next() != ";":
          pass
      else:
        self.parse_field(msg, tok, scope, known)
    self.expect("}")

  def parse_field(self, msg, first: str, scope: List[str], known: dict):
    label = _F.LABEL_OPTIONAL
    if first == "repeated":
      label = _F.LABEL_REPEATED
      first = self.next()
    elif first == "optional":
      first = self.next()
    if first == "map":
      self.expect("<")
      ktype = self.next()
      self.expect(",")
      vtype = self.next()
      self.expect(">")
      name = self.next()
      self.expect("=")
      number = int(self.next())
      self.skip_options()
      self.expect(";")
      entry = msg.nested_type.add()
      entry.name = "".join(p.capitalize() for p in name.split("_")) + "Entry"
      entry.options.map_entry = True
      kf = entry.field.add()
      kf.name, kf.number, kf.label = "key", 1, _F.LABEL_OPTIONAL
      self._set_type(kf, ktype, scope, known)
      vf = entry.field.add()
      vf.name, vf.number, vf.label = "value", 2, _F.LABEL_OPTIONAL
      self._set_type(vf, vtype, scope, known)
      f = msg.field.add()
      f.name, f.number, f.label = name, number, _F.LABEL_REPEATED
      f.type = _F.TYPE_MESSAGE
      f.type_name = "." + self.package + "." + ".".join(scope + [entry.name])
      return f
    type_name = first
    name = self.next()
    self.expect("=")
    number = int(self.next())
    self.skip_options()
    self.expect(";")
    f = msg.field.add()
    f.name, f.number, f.label = name, number, label
    self._set_type(f, type_name, scope, known)
    return f

  def parse_file(self, file_proto) -> None:
    known = None
    while self.i < len(self.t):
      tok = self.next()
      if tok == "syntax":
        self.expect("=")
        file_proto.syntax = self.next().strip('"')
        self.expect(";")
      elif tok == "package":
        self.package = self.next()
        file_proto.package = self.package
        self.expect(";")
      elif tok == "import":
        dep = self.next().strip('"')
        if dep == "public":
          dep = self.next().strip('"')
        file_proto.dependency.append(dep)
        self.expect(";")
      elif tok == "option":
        while self.next() != ";":
          pass
      elif tok == "message":
        if known is None:
          known = self.collect()
        self.parse_message(file_proto.message_type.add(), [], known)
      elif tok == "enum":
        self.parse_enum(file_proto.enum_type.add())
      else:
        raise ValueError(f"proto parse: unexpected top-level token {tok!r}")


def build_module(proto_path: str, module_name: str, virtual_name: str) -> types.ModuleType:
  """Parses ``proto_path`` and returns a module exposing its top-level messages."""
  with open(proto_path, "r", encoding="utf-8") as fh:
    text = fh.read()
  file_proto = descriptor_pb2.FileDescriptorProto()
  file_proto.name = virtual_name
  _Parser(_tokenize(text)).parse_file(file_proto)
  pool = descriptor_pool.Default()
  try:
    file_desc = pool.Add(file_proto)
  except TypeError:  # some runtimes only offer AddSerializedFile
    file_desc = pool.AddSerializedFile(file_proto.SerializeToString())
  if file_desc is None:
    file_desc = pool.FindFileByName(virtual_name)
  mod = types.ModuleType(module_name)
  for name, desc in file_desc.message_types_by_name.items():
    setattr(mod, name, message_factory.GetMessageClass(desc))
  mod.DESCRIPTOR = file_desc
  return mod
