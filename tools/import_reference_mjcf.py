#!/usr/bin/env python3
"""Import a reference MJCF humanoid into the compact body table this repo ships.

The reference's packaged fixture model (reference
smpl_sim/data/assets/mjcf/smpl_humanoid.xml, used when the licensed SMPL files
are absent: smpl_sim/envs/humanoid_env.py:249-254) is the *input data* of the
hot path.  /root/reference does not exist on the GPU box, and reference source
files must not be copied, so this tool extracts only the numbers that define
the model (tree, offsets, geom shapes, densities, joint ranges, excludes) into
a compact JSON table.  `smplsim_amd.mjcf_writer.table_to_mjcf` regenerates an
equivalent MJCF string from the table; tests/test_mjcf.py checks (when
/root/reference is present) that compiling the regenerated XML gives exactly
the same model constants as compiling the reference XML.

Usage (run in the build container, where /root/reference exists):
    python tools/import_reference_mjcf.py \
        /root/reference/smpl_sim/data/assets/mjcf/smpl_humanoid.xml \
        smplsim_amd/data/smpl_humanoid.json
"""
import json
import sys
import xml.etree.ElementTree as ET


def _floats(s):
    return [float(x) for x in s.split()]


def import_mjcf(path):
    root = ET.parse(path).getroot()
    dflt = root.find("default")
    table = {
        "model": root.get("model", "humanoid"),
        "default_joint": dict(dflt.find("joint").attrib) if dflt is not None and dflt.find("joint") is not None else {},
        "default_geom": {k: v for k, v in (dflt.find("geom").attrib.items() if dflt is not None and dflt.find("geom") is not None else []) if k != "rgba"},
        "floor": None,
        "bodies": [],
        "excludes": [],
        "vel_sensors": False,
    }
    wb = root.find("worldbody")
    for g in wb.findall("geom"):
        if g.get("type") == "plane":
            table["floor"] = {"name": g.get("name"), "pos": _floats(g.get("pos", "0 0 0")),
                              "size": _floats(g.get("size")),
                              "conaffinity": g.get("conaffinity"), "condim": g.get("condim")}

    def walk(elem, parent_name):
        for b in elem.findall("body"):
            entry = {"name": b.get("name"), "parent": parent_name, "pos": _floats(b.get("pos", "0 0 0"))}
            if b.get("quat") is not None:
                entry["quat"] = _floats(b.get("quat"))
            entry["freejoint"] = b.find("freejoint") is not None
            joints = []
            for j in b.findall("joint"):
                jd = {"name": j.get("name"), "axis": _floats(j.get("axis")),
                      "range": _floats(j.get("range")) if j.get("range") else None}
                for k in ("armature", "damping", "stiffness", "type", "pos", "user"):
                    if j.get(k) is not None:
                        jd[k] = j.get(k)
                joints.append(jd)
            entry["joints"] = joints
            geoms = []
            for g in b.findall("geom"):
                gd = {"name": g.get("name"), "type": g.get("type")}
                for k in ("pos", "size", "quat", "fromto"):
                    if g.get(k) is not None:
                        gd[k] = _floats(g.get(k))
                for k in ("density", "contype", "conaffinity"):
                    if g.get(k) is not None:
                        gd[k] = g.get(k)
                geoms.append(gd)
            entry["geoms"] = geoms
            table["bodies"].append(entry)
            walk(b, b.get("name"))

    walk(wb, None)
    act = root.find("actuator")
    table["motors"] = [{"name": m.get("name"), "joint": m.get("joint"), "gear": m.get("gear", "1")}
                       for m in (act.findall("motor") if act is not None else [])]
    con = root.find("contact")
    if con is not None:
        table["excludes"] = [[e.get("body1"), e.get("body2")] for e in con.findall("exclude")]
    sen = root.find("sensor")
    table["vel_sensors"] = sen is not None and len(sen.findall("framelinvel")) > 0
    return table


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    t = import_mjcf(src)
    with open(dst, "w") as f:
        json.dump(t, f, separators=(",", ":"))
    print(f"{dst}: {len(t['bodies'])} bodies, {len(t['motors'])} motors, {len(t['excludes'])} excludes")
