"""Reads the reference's real-data scene data/sfm/fountain11.bin (Strecha fountain-P11: a cereal portable-binary
theia::Reconstruction, written by reconstruction_writer.cc; layout from the serialize() members of reconstruction.h:183,
view.h:119, track.h:95, feature.h:110, camera/camera.h:211, camera_intrinsics_prior.h:76,118, io/eigen_serializable.h)
and the ground-truth cameras gt_fountain11.bin, in THIS container only, and commits the parsed arrays as
tests/golden/fountain11.npz (data, not source: cameras, intrinsics, tracks, observations)."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/data/sfm"


class Reader:
    def __init__(self, buf):
        self.b = buf; self.p = 0
        self.versions = {}
        self.poly_names = {}
        self.shared = {}
        self.endian = self.u8()
        assert self.endian == 1          # little endian archive

    def raw(self, n):
        v = self.b[self.p:self.p + n]; self.p += n
        assert len(v) == n
        return v

    def u8(self): return self.raw(1)[0]
    def boolean(self): return bool(self.u8())
    def u32(self): return struct.unpack("<I", self.raw(4))[0]
    def i32(self): return struct.unpack("<i", self.raw(4))[0]
    def u64(self): return struct.unpack("<Q", self.raw(8))[0]
    def f64(self): return struct.unpack("<d", self.raw(8))[0]
    def doubles(self, n): return np.frombuffer(self.raw(8 * n), dtype="<f8").copy()
    def string(self): return self.raw(self.u64()).decode()

    def version(self, cls):
        if cls not in self.versions:
            self.versions[cls] = self.u32()
        return self.versions[cls]

    def eigen(self, dtype="<f8", size=8):
        r, c = self.i32(), self.i32()
        return np.frombuffer(self.raw(size * r * c), dtype=dtype).copy().reshape(c, r).T   # column-major


def prior(rd, n):
    rd.version(f"Prior<{n}>")
    return rd.boolean(), rd.doubles(n)


def intrinsics_prior(rd):
    v = rd.version("CameraIntrinsicsPrior")
    if v == 0:        # camera_intrinsics_prior.h:166-183: focal, ppx, ppy, aspect, skew, rd1, rd2
        return {name: prior(rd, 1) for name in ("focal_length", "ppx", "ppy", "aspect_ratio", "skew", "rd1", "rd2")}
    assert v >= 4, v
    out = {"image_width": rd.i32(), "image_height": rd.i32(), "model": rd.string()}
    for name, n in (("focal_length", 1), ("principal_point", 2), ("aspect_ratio", 1), ("skew", 1), ("radial_distortion", 4),
                    ("tangential_distortion", 2), ("position", 3), ("orientation", 3), ("latitude", 1), ("longitude", 1),
                    ("altitude", 1)):
        out[name] = prior(rd, n)
    return out


def camera(rd):
    v = rd.version("Camera")
    if v == 0:       # camera.h:217-248: pinhole only, extrinsics + intrinsics as one block
        p = rd.doubles(6 + 7)
        return p[:6], "theia::PinholeCameraModel", p[6:], (rd.i32(), rd.i32()), -1
    ext = rd.doubles(6)
    # std::shared_ptr<CameraIntrinsicsModel>, polymorphic
    pid = rd.u32()
    if pid & 0x80000000:
        rd.poly_names[pid & 0x7fffffff] = rd.string()
    name = rd.poly_names[pid & 0x7fffffff]
    sid = rd.u32()
    if sid & 0x80000000:          # first occurrence of this shared object: its data follows
        mv = rd.version(name)
        assert mv >= 1, (name, mv)
        rd.version("CameraIntrinsicsModel")
        rd.shared[sid & 0x7fffffff] = rd.doubles(rd.u64())
    params = rd.shared[sid & 0x7fffffff]
    size = (rd.i32(), rd.i32())
    return ext, name, params, size, sid & 0x7fffffff


def view(rd):
    """The file predates the timestamp / covariance / prior members of today's headers (class versions 0 in both): a View
    is name, is_estimated, camera, intrinsics prior and the TrackId -> Vector2d feature map."""
    rd.version("View")
    out = {"name": rd.string(), "is_estimated": rd.boolean()}
    out["camera"] = camera(rd)
    out["prior"] = intrinsics_prior(rd)
    feats = {}
    for _ in range(rd.u64()):
        tid = rd.u32(); feats[tid] = rd.eigen().reshape(-1)
    out["features"] = feats
    return out


def track(rd):
    rd.version("Track")
    return {"is_estimated": rd.boolean(), "view_ids": [rd.u32() for _ in range(rd.u64())], "point": rd.eigen().reshape(-1),
            "color": rd.eigen("u1", 1).reshape(-1)}


def reconstruction(path):
    rd = Reader(open(path, "rb").read())
    rd.version("Reconstruction")
    rd.u32(); rd.u32()                                        # next track / view id
    for _ in range(rd.u64()): rd.string(); rd.u32()            # view name -> id
    views = {}
    for _ in range(rd.u64()):
        vid = rd.u32(); views[vid] = view(rd)
    tracks = {}
    for _ in range(rd.u64()):
        tid = rd.u32(); tracks[tid] = track(rd)
    groups = {}
    if rd.p < len(rd.b):
        for _ in range(rd.u64()):
            vid = rd.u32(); groups[vid] = rd.u32()
        for _ in range(rd.u64()):
            rd.u32()
            for _ in range(rd.u64()): rd.u32()
    assert rd.p == len(rd.b), (rd.p, len(rd.b))
    return views, tracks, groups, rd.versions


def main():
    views, tracks, groups, versions = reconstruction(os.path.join(SRC, "fountain11.bin"))
    gviews, gtracks, _, _ = reconstruction(os.path.join(SRC, "gt_fountain11.bin"))
    print("class versions in the file:", versions)
    vids = sorted(views)
    est_tracks = sorted(t for t in tracks if tracks[t]["is_estimated"])
    tindex = {t: i for i, t in enumerate(est_tracks)}
    names = [views[v]["name"] for v in vids]
    models = sorted({views[v]["camera"][1] for v in vids})
    print(len(vids), "views", len(tracks), "tracks", len(est_tracks), "estimated; models", models)
    cam_ext = np.array([views[v]["camera"][0] for v in vids])
    npar = max(len(views[v]["camera"][2]) for v in vids)
    intr = np.zeros((len(vids), npar))
    for i, v in enumerate(vids): intr[i, :len(views[v]["camera"][2])] = views[v]["camera"][2]
    size = np.array([views[v]["camera"][3] for v in vids])
    oc, ot, ouv = [], [], []
    for i, v in enumerate(vids):
        for t, f in sorted(views[v]["features"].items()):
            if t in tindex:
                oc.append(i); ot.append(tindex[t]); ouv.append(f[:2])
    pts = np.array([tracks[t]["point"] for t in est_tracks])
    gnames = {gviews[v]["name"]: gviews[v]["camera"][0] for v in gviews}
    gt_ext = np.array([gnames.get(n, np.full(6, np.nan)) for n in names])
    np.savez_compressed(os.path.join(HERE, "fountain11.npz"),
                        view_names=np.array(names), view_estimated=np.array([views[v]["is_estimated"] for v in vids]),
                        cam_ext=cam_ext, intrinsics=intr, intrinsics_model=np.array([views[v]["camera"][1] for v in vids]),
                        image_size=size, intrinsics_group=np.array([groups.get(v, -1) for v in vids]),
                        obs_cam=np.array(oc, dtype=np.int32), obs_track=np.array(ot, dtype=np.int32), obs_uv=np.array(ouv),
                        points=pts, gt_cam_ext=gt_ext)
    print("observations", len(oc), "track length max", np.bincount(np.array(ot)).max())


if __name__ == "__main__":
    sys.exit(main())
