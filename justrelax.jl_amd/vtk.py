"""save_vtk: vertex and cell data of a rectilinear grid as one VTK XML file (.vtr) plus an optional ParaView collection (.pvd).

Reference: src/IO/VTK.jl:10-64 (`pack_velocity`, `add_field!`) and :132-178 (`save_vtk(fname, xvi, xci, data_v, data_c, velocity; precision, t, pvd)`), which
go through WriteVTK.jl; here the XML is written directly (ASCII-free: base64 "binary" data arrays, little endian, UInt32 headers).  Host-side I/O around the
hot path: device arrays are brought to the host with `Array_` first.
"""
from __future__ import annotations

import base64
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

from .convert import Array_

_VTK_TYPE = {np.dtype("float32"): "Float32", np.dtype("float64"): "Float64"}


def _b64(a: np.ndarray) -> str:
    raw = np.ascontiguousarray(a).tobytes()
    return (base64.b64encode(struct.pack("<I", len(raw))) + base64.b64encode(raw)).decode("ascii")


def _data_array(parent, name, a: np.ndarray, ncomp=1):
    e = ET.SubElement(parent, "DataArray", type=_VTK_TYPE[a.dtype], Name=name, NumberOfComponents=str(ncomp), format="binary")
    e.text = _b64(a)
    return e


def pack_velocity(velocity, precision):
    """(3, size...) array of the velocity components; missing components are zeros (VTK.jl:22-37)"""
    if len(velocity) > 3:
        raise ValueError(f"velocity must have at most 3 components, got {len(velocity)}")
    v0 = np.asarray(Array_(velocity[0]))
    out = np.zeros((3,) + v0.shape, dtype=precision)
    for i, v in enumerate(velocity):
        v = np.asarray(Array_(v))
        if v.shape != v0.shape:
            raise ValueError(f"velocity components must share their axes: {v.shape} vs {v0.shape}")
        out[i] = v.astype(precision)
    return out


def save_vtk(fname, xvi, xci, data_v: dict, data_c: dict, velocity, *, precision=np.float32, t=0, pvd=None):
    """save_vtk(fname, xvi, xci, data_v, data_c, velocity; precision = Float32, t = 0, pvd = nothing) -- VTK.jl:132-178.  One file `fname.vtr` on the grid
    spanned by the vertices: fields of the vertices' size go to the point data, fields of the cells' size to the cell data (whichever dict they came in),
    `Velocity` (given on the vertices) is a 3-component point vector, `TimeValue` a field datum; `pvd` appends the file to that ParaView collection."""
    nv, nc = tuple(len(x) for x in xvi), tuple(len(x) for x in xci)
    if nv != tuple(n + 1 for n in nc):
        raise ValueError(f"the vertex grid must have one node more per dimension than the center grid: {nv} vs {nc}")
    vel = pack_velocity(velocity, precision)
    if vel.shape[1:] != nv:
        raise ValueError(f"velocity must be given on the vertices: {vel.shape[1:]} vs {nv}")
    nd = len(nv)
    ext = " ".join(f"0 {n - 1}" for n in nv) + " 0 0" * (3 - nd)
    root = ET.Element("VTKFile", type="RectilinearGrid", version="1.0", byte_order="LittleEndian", header_type="UInt32")
    grid = ET.SubElement(root, "RectilinearGrid", WholeExtent=ext)
    if t is not None:
        fd = ET.SubElement(grid, "FieldData")
        e = ET.SubElement(fd, "DataArray", type="Float64", Name="TimeValue", NumberOfTuples="1", format="binary")
        e.text = _b64(np.array([float(t)], dtype=np.float64))
    piece = ET.SubElement(grid, "Piece", Extent=ext)
    pdata, cdata = ET.SubElement(piece, "PointData", Vectors="Velocity"), ET.SubElement(piece, "CellData")
    for name, arr in list(data_v.items()) + list(data_c.items()):          # add_field!, VTK.jl:46-64
        a = np.asarray(Array_(arr)).astype(precision)
        if a.shape == nv:
            _data_array(pdata, str(name), a.ravel(order="F"))
        elif a.shape == nc:
            _data_array(cdata, str(name), a.ravel(order="F"))
        else:
            raise ValueError(f"{name} has size {a.shape}, which matches neither the {nv} vertices nor the {nc} cells of the grid")
    # tuples are contiguous in VTK: (3, nx, ny, nz) in column-major order is exactly component-fastest
    _data_array(pdata, "Velocity", vel.ravel(order="F"), ncomp=3)
    coords = ET.SubElement(piece, "Coordinates")
    for d, nm in enumerate("xyz"):
        c = np.asarray(xvi[d], dtype=np.float64) if d < nd else np.zeros(1)
        _data_array(coords, f"{nm}_coordinates", c)
    out = f"{fname}.vtr"
    ET.ElementTree(root).write(out, xml_declaration=True, encoding="utf-8")
    if pvd is not None:
        _append_pvd(f"{pvd}.pvd", out, t)
    return out


def _append_pvd(pvd_file, vtk_file, t):
    """paraview_collection(pvd; append = true) + collection_add_timestep: create the collection or add one DataSet entry to it"""
    if os.path.exists(pvd_file):
        tree = ET.parse(pvd_file)
        coll = tree.getroot().find("Collection")
    else:
        r = ET.Element("VTKFile", type="Collection", version="1.0", byte_order="LittleEndian")
        coll = ET.SubElement(r, "Collection")
        tree = ET.ElementTree(r)
    rel = os.path.relpath(vtk_file, os.path.dirname(os.path.abspath(pvd_file)))
    ET.SubElement(coll, "DataSet", timestep=repr(float(0 if t is None else t)), part="0", file=rel)
    tree.write(pvd_file, xml_declaration=True, encoding="utf-8")
