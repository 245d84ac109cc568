"""Regenerate an MJCF string from the compact body table shipped in smplsim_amd/data.

The reference falls back to a packaged MJCF file when the licensed SMPL model
files are absent (reference smpl_sim/envs/humanoid_env.py:249-254).  This repo
ships the same model as a compact JSON table (see tools/import_reference_mjcf.py)
and writes the MJCF text from it, so `HumanoidEnv.default_xml_str` exists here too.
"""
import json
from importlib.resources import files


def _fmt(vals):
    return " ".join(repr(float(v)) for v in vals)


def table_to_mjcf(table):
    out = []
    w = out.append
    w(f'<mujoco model="{table.get("model", "humanoid")}">')
    w('  <compiler coordinate="local"/>')
    w("  <default>")
    dj = " ".join(f'{k}="{v}"' for k, v in table["default_joint"].items())
    dg = " ".join(f'{k}="{v}"' for k, v in table["default_geom"].items())
    w(f"    <joint {dj}/>")
    w(f"    <geom {dg}/>")
    w("  </default>")
    w("  <worldbody>")
    fl = table["floor"]
    if fl is not None:
        w(f'    <geom name="{fl["name"]}" type="plane" pos="{_fmt(fl["pos"])}" size="{_fmt(fl["size"])}" '
          f'conaffinity="{fl["conaffinity"]}" condim="{fl["condim"]}"/>')
    children = {}
    for b in table["bodies"]:
        children.setdefault(b["parent"], []).append(b)

    def emit(b, ind):
        pad = " " * ind
        q = f' quat="{_fmt(b["quat"])}"' if "quat" in b else ""
        w(f'{pad}<body name="{b["name"]}" pos="{_fmt(b["pos"])}"{q}>')
        if b["freejoint"]:
            w(f'{pad}  <freejoint name="{b["name"]}"/>')
        for j in b["joints"]:
            extra = "".join(f' {k}="{j[k]}"' for k in ("type", "pos", "armature", "damping", "stiffness", "user") if k in j)
            rng = f' range="{_fmt(j["range"])}"' if j.get("range") is not None else ""
            w(f'{pad}  <joint name="{j["name"]}" axis="{_fmt(j["axis"])}"{rng}{extra}/>')
        for g in b["geoms"]:
            attrs = f'name="{g["name"]}" type="{g["type"]}"'
            for k in ("pos", "size", "quat", "fromto"):
                if k in g:
                    attrs += f' {k}="{_fmt(g[k])}"'
            for k in ("density", "contype", "conaffinity"):
                if k in g:
                    attrs += f' {k}="{g[k]}"'
            w(f"{pad}  <geom {attrs}/>")
        for c in children.get(b["name"], []):
            emit(c, ind + 2)
        w(f"{pad}</body>")

    for b in children.get(None, []):
        emit(b, 4)
    w("  </worldbody>")
    w("  <actuator>")
    for m in table["motors"]:
        w(f'    <motor name="{m["name"]}" joint="{m["joint"]}" gear="{m["gear"]}"/>')
    w("  </actuator>")
    if table["excludes"]:
        w("  <contact>")
        for a, b in table["excludes"]:
            w(f'    <exclude body1="{a}" body2="{b}"/>')
        w("  </contact>")
    if table.get("vel_sensors"):
        w("  <sensor>")
        for kind in ("framelinvel", "frameangvel"):
            for b in table["bodies"]:
                w(f'    <{kind} name="sensor_{b["name"]}_{kind}" objtype="xbody" objname="{b["name"]}"/>')
        w("  </sensor>")
    w("</mujoco>")
    return "\n".join(out) + "\n"


def load_table(name="smpl_humanoid"):
    """name: 'smpl_humanoid' (24 bodies, nv=75) or 'smplx_humanoid' (52 bodies, nv=159)."""
    with files("smplsim_amd").joinpath(f"data/{name}.json").open("r") as f:
        return json.load(f)


def default_xml_str(name="smpl_humanoid"):
    return table_to_mjcf(load_table(name))


def scaled_xml_str(name="smpl_humanoid", scale=1.0, limb_scale=None):
    """A body-shape variant of a packaged model: every body offset, geom position / size / fromto scaled by `scale`, times
    limb_scale[body] for that body's own offset and geoms (stand-in for the MJCFs SMPL_Robot writes for different betas,
    reference smpl_sim/smpllib/smpl_local_robot.py — those need the SMPL model files)."""
    import copy
    t = copy.deepcopy(load_table(name))
    limb_scale = limb_scale or {}
    for b in t["bodies"]:
        f = scale * limb_scale.get(b["name"], 1.0)
        if b["parent"] is not None:
            b["pos"] = [f * x for x in b["pos"]]
        for g in b["geoms"]:
            for k in ("pos", "size", "fromto"):
                if k in g:
                    g[k] = [f * x for x in g[k]]
    return table_to_mjcf(t)
