"""ROS-free reader / writer for rosbag v2.0 files carrying the reference's person_msgs.

The demo of the reference replays `poses2D_16cam.bag` (per-camera `person_msgs/Person2DList` topics) into skeleton_3d and records
`person_msgs/PersonCovList` outputs.  This module ingests / produces such bags without ROS so that recorded streams can be fed to the
replay harness (harness.py / host/se3d_replay) and synthetic scenes can be exported for the reference nodes:

  read_bag(path)                      -> connections + (topic, record time, payload) of every message (chunks: none, bz2)
  read_person2d_stream(path)          -> {topic: [CameraMsg]} in bag order
  write_bag(path, messages)           -> a valid v2.0 bag (uncompressed chunks, connection / index / chunk-info records)
  (de)serialize_person2d_list, (de)serialize_person_cov_list: the ROS1 wire format of person_msgs/msg/*.msg

Format: ROS bag format 2.0 (record = header length, header fields "name=value" each with a 4-byte length, data length, data; op codes
0x03 bag header, 0x05 chunk, 0x07 connection, 0x02 message data, 0x04 index data, 0x06 chunk info).  Wire format: little endian,
string / variable array = uint32 count + elements, fixed arrays without count, time = uint32 secs + uint32 nsecs.
No bag of the reference is available in this environment (README.md:41-45 points to an external download), so the reader is checked
against the writer and against hand-built records only ("parity unpinned"); message md5 sums are computed with the genmsg
algorithm from the reference's .msg texts and the well-known std_msgs / geometry_msgs sums.
"""
import bz2
import hashlib
import struct

import numpy as np

from . import person_msgs as pm

MAGIC = b"#ROSBAG V2.0\n"
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 2, 3, 4, 5, 6, 7

# person_msgs/msg/*.msg with comments removed (field order = wire order)
MSG_DEFS = {
    "person_msgs/Keypoint2D": "float32 x\nfloat32 y\nfloat32 score\nfloat32[3] cov",
    "person_msgs/Person2D": "float32 score\nKeypoint2D[] keypoints\nfloat32[4] bbox",
    "person_msgs/Person2DList": "std_msgs/Header header\nfloat32 fb_delay\nPerson2D[] persons",
    "person_msgs/KeypointWithCovariance": "geometry_msgs/Point joint\nfloat32 score\nfloat64[6] cov",
    "person_msgs/PersonCov": "uint32 id\nfloat32 score\nKeypointWithCovariance[] keypoints\ngeometry_msgs/Pose bbox_center\ngeometry_msgs/Vector3 bbox_size",
    "person_msgs/PersonCovList": "std_msgs/Header header\ntime[] ts_per_cam\nfloat32[] fb_delay_per_cam\nPersonCov[] persons",
}
KNOWN_MD5 = {"std_msgs/Header": "2176decaecbce78abc3b96ef049fabed", "geometry_msgs/Point": "4a842b65f413084dc2b10fb484ea7f17",
             "geometry_msgs/Vector3": "4a842b65f413084dc2b10fb484ea7f17", "geometry_msgs/Quaternion": "a779879fadf0160734f906b8c19c7004",
             "geometry_msgs/Pose": "e45d45a5a1ce597b249e23fb30fc871f"}
BUILTIN = {"bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "string", "time", "duration", "char", "byte"}


def msg_md5(type_name):
    """genmsg md5: fields of non-builtin type are replaced by the md5 of that type (array suffix dropped)."""
    if type_name in KNOWN_MD5:
        return KNOWN_MD5[type_name]
    lines = []
    for line in MSG_DEFS[type_name].split("\n"):
        ftype, fname = line.split()
        base = ftype.split("[")[0]
        if base in BUILTIN:
            lines.append("%s %s" % (ftype, fname))
        else:
            full = base if "/" in base else "person_msgs/" + base
            lines.append("%s %s" % (msg_md5(full), fname))
    return hashlib.md5("\n".join(lines).encode()).hexdigest()


# ---------------------------------------------------------------- wire format of the messages
def _header(seq, stamp_ns, frame_id):
    fid = frame_id.encode()
    return struct.pack("<III", seq, stamp_ns // 1000000000, stamp_ns % 1000000000) + struct.pack("<I", len(fid)) + fid


def _read_header(buf, pos):
    seq, secs, nsecs, n = struct.unpack_from("<IIII", buf, pos)
    pos += 16
    return seq, secs * 1000000000 + nsecs, bytes(buf[pos:pos + n]).decode(), pos + n


def serialize_person2d_list(stamp_ns, persons, fb_delay=-1.0, frame_id="", seq=0):
    """persons: array of person_msgs.person2d_dt (17 keypoints each)."""
    out = [_header(seq, stamp_ns, frame_id), struct.pack("<fI", fb_delay, len(persons))]
    for p in persons:
        out.append(struct.pack("<fI", float(p["score"]), pm.NUM_KEYPOINTS))
        kp = p["keypoints"]
        rec = np.zeros((pm.NUM_KEYPOINTS, 6), "<f4")
        rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3:] = kp["x"], kp["y"], kp["score"], kp["cov"]
        out.append(rec.tobytes())
        out.append(np.asarray(p["bbox"], "<f4").tobytes())
    return b"".join(out)


def deserialize_person2d_list(buf):
    """-> dict(seq, stamp_ns, frame_id, fb_delay, persons[person2d_dt], skipped).  Persons whose keypoint count is not 17 are skipped like
    the node does (S3D:319-320) and counted in `skipped`."""
    seq, stamp, frame_id, pos = _read_header(buf, 0)
    fb_delay, n = struct.unpack_from("<fI", buf, pos)
    pos += 8
    persons, skipped = [], 0
    for _ in range(n):
        score, nk = struct.unpack_from("<fI", buf, pos)
        pos += 8
        rec = np.frombuffer(buf, "<f4", nk * 6, pos).reshape(nk, 6)
        pos += nk * 24
        bbox = np.frombuffer(buf, "<f4", 4, pos)
        pos += 16
        if nk != pm.NUM_KEYPOINTS:
            skipped += 1
            continue
        p = np.zeros((), pm.person2d_dt)
        p["score"], p["bbox"] = score, bbox
        p["keypoints"]["x"], p["keypoints"]["y"], p["keypoints"]["score"], p["keypoints"]["cov"] = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3:]
        persons.append(p)
    assert pos == len(buf), "trailing bytes in Person2DList"
    return dict(seq=seq, stamp_ns=stamp, frame_id=frame_id, fb_delay=float(fb_delay), skipped=skipped,
                persons=np.array(persons, pm.person2d_dt) if persons else np.zeros(0, pm.person2d_dt))


def serialize_person_cov_list(stamp_ns, persons, ts_per_cam_ns=(), fb_delay_per_cam=(), frame_id="base", seq=0):
    """persons: array of person_msgs.person3d_dt (21 keypoints); bbox_center / bbox_size are written as identity pose / zeros."""
    out = [_header(seq, stamp_ns, frame_id), struct.pack("<I", len(ts_per_cam_ns))]
    out += [struct.pack("<II", t // 1000000000, t % 1000000000) for t in ts_per_cam_ns]
    out.append(struct.pack("<I", len(fb_delay_per_cam)) + np.asarray(fb_delay_per_cam, "<f4").tobytes())
    out.append(struct.pack("<I", len(persons)))
    for p in persons:
        out.append(struct.pack("<IfI", int(p["id"]), float(p["score"]), pm.NUM_FUSION_KEYPOINTS))
        for k in p["keypoints"]:
            out.append(np.asarray(k["joint"], "<f8").tobytes() + struct.pack("<f", float(k["score"])) + np.asarray(k["cov"], "<f8").tobytes())
        out.append(struct.pack("<7d", 0, 0, 0, 0, 0, 0, 1) + struct.pack("<3d", 0, 0, 0))
    return b"".join(out)


def deserialize_person_cov_list(buf):
    seq, stamp, frame_id, pos = _read_header(buf, 0)
    (n,) = struct.unpack_from("<I", buf, pos)
    pos += 4
    ts = [s * 1000000000 + ns for s, ns in struct.iter_unpack("<II", bytes(buf[pos:pos + 8 * n]))]
    pos += 8 * n
    (n,) = struct.unpack_from("<I", buf, pos)
    pos += 4
    fb = np.frombuffer(buf, "<f4", n, pos).copy()
    pos += 4 * n
    (n,) = struct.unpack_from("<I", buf, pos)
    pos += 4
    persons = np.zeros(n, pm.person3d_dt)
    for i in range(n):
        pid, score, nk = struct.unpack_from("<IfI", buf, pos)
        pos += 12
        persons[i]["id"], persons[i]["score"] = pid, score
        for k in range(nk):
            v = struct.unpack_from("<3df6d", buf, pos)
            pos += 76
            if k < pm.NUM_FUSION_KEYPOINTS:
                persons[i]["keypoints"][k]["joint"], persons[i]["keypoints"][k]["score"], persons[i]["keypoints"][k]["cov"] = v[:3], v[3], v[4:]
        pos += 80  # bbox_center (Pose: 7 float64) + bbox_size (Vector3: 3 float64)
    assert pos == len(buf), "trailing bytes in PersonCovList"
    return dict(seq=seq, stamp_ns=stamp, frame_id=frame_id, ts_per_cam_ns=ts, fb_delay_per_cam=fb, persons=persons)


# ---------------------------------------------------------------- bag records
def _fields(d):
    out = []
    for k, v in d.items():
        f = k.encode() + b"=" + v
        out.append(struct.pack("<I", len(f)) + f)
    return b"".join(out)


def _record(header, data):
    h = _fields(header)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _parse_fields(buf):
    d, pos = {}, 0
    while pos < len(buf):
        (n,) = struct.unpack_from("<I", buf, pos)
        f = bytes(buf[pos + 4:pos + 4 + n])
        k, v = f.split(b"=", 1)
        d[k.decode()] = v
        pos += 4 + n
    return d


def _records(buf, pos, end):
    while pos < end:
        (hl,) = struct.unpack_from("<I", buf, pos)
        hdr = _parse_fields(buf[pos + 4:pos + 4 + hl])
        pos += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, pos)
        yield hdr, buf[pos + 4:pos + 4 + dl]
        pos += 4 + dl


def read_bag(path):
    """-> (connections {id: dict(topic, type, md5sum)}, [(topic, time_ns, payload bytes)]) in file order."""
    buf = memoryview(open(path, "rb").read())
    if bytes(buf[:len(MAGIC)]) != MAGIC:
        raise ValueError("%s is not a rosbag v2.0 file" % path)
    conns, msgs = {}, []

    def handle(hdr, data):
        op = hdr["op"][0]
        if op == OP_CONNECTION:
            cid = struct.unpack("<I", hdr["conn"])[0]
            info = _parse_fields(data)
            conns[cid] = dict(topic=hdr["topic"].decode(), type=info.get("type", b"").decode(), md5sum=info.get("md5sum", b"").decode())
        elif op == OP_MSG:
            cid = struct.unpack("<I", hdr["conn"])[0]
            secs, nsecs = struct.unpack("<II", hdr["time"])
            msgs.append((cid, secs * 1000000000 + nsecs, bytes(data)))

    for hdr, data in _records(buf, len(MAGIC), len(buf)):
        op = hdr["op"][0]
        if op == OP_CHUNK:
            comp = hdr["compression"].decode()
            if comp == "none":
                inner = data
            elif comp == "bz2":
                inner = memoryview(bz2.decompress(bytes(data)))
            else:
                raise NotImplementedError("chunk compression %r (only none / bz2; re-record with `rosbag compress --bz2` or uncompressed)" % comp)
            for h2, d2 in _records(inner, 0, len(inner)):
                handle(h2, d2)
        else:
            handle(hdr, data)  # connection records after the chunks repeat the in-chunk ones; index / chunk-info records are not needed
    return conns, [(conns[c]["topic"], t, payload) for c, t, payload in msgs]


def write_bag(path, messages, chunk_messages=64):
    """messages: iterable of (topic, type_name, time_ns, payload bytes) in recording order."""
    conn_ids, chunks, cur, cur_conns, cur_index = {}, [], [], set(), {}
    conn_recs = {}

    def conn_record(cid, topic, tname):
        info = _fields({"topic": topic.encode(), "type": tname.encode(), "md5sum": msg_md5(tname).encode(),
                        "message_definition": MSG_DEFS.get(tname, "").encode()})
        return _record({"op": bytes([OP_CONNECTION]), "conn": struct.pack("<I", cid), "topic": topic.encode()}, info)

    def flush():
        nonlocal cur, cur_conns, cur_index
        if cur:
            chunks.append((b"".join(cur), cur_index))
            cur, cur_conns, cur_index = [], set(), {}

    off = 0
    for topic, tname, t, payload in messages:
        if topic not in conn_ids:
            conn_ids[topic] = len(conn_ids)
            conn_recs[conn_ids[topic]] = conn_record(conn_ids[topic], topic, tname)
        cid = conn_ids[topic]
        if not cur:
            off = 0
        if cid not in cur_conns:
            cur.append(conn_recs[cid])
            off += len(conn_recs[cid])
            cur_conns.add(cid)
        rec = _record({"op": bytes([OP_MSG]), "conn": struct.pack("<I", cid), "time": struct.pack("<II", t // 1000000000, t % 1000000000)}, payload)
        cur_index.setdefault(cid, []).append((t, off))
        cur.append(rec)
        off += len(rec)
        if sum(len(v) for v in cur_index.values()) >= chunk_messages:
            flush()
    flush()
    body, chunk_infos = [], []
    pos = len(MAGIC) + 4096
    for data, index in chunks:
        rec = _record({"op": bytes([OP_CHUNK]), "compression": b"none", "size": struct.pack("<I", len(data))}, data)
        chunk_pos = pos
        body.append(rec)
        pos += len(rec)
        for cid, entries in index.items():
            irec = _record({"op": bytes([OP_INDEX]), "ver": struct.pack("<I", 1), "conn": struct.pack("<I", cid), "count": struct.pack("<I", len(entries))},
                           b"".join(struct.pack("<III", t // 1000000000, t % 1000000000, o) for t, o in entries))
            body.append(irec)
            pos += len(irec)
        times = [t for e in index.values() for t, _ in e]
        chunk_infos.append(_record({"op": bytes([OP_CHUNK_INFO]), "ver": struct.pack("<I", 1), "chunk_pos": struct.pack("<Q", chunk_pos),
                                    "start_time": struct.pack("<II", min(times) // 1000000000, min(times) % 1000000000),
                                    "end_time": struct.pack("<II", max(times) // 1000000000, max(times) % 1000000000),
                                    "count": struct.pack("<I", len(index))},
                                   b"".join(struct.pack("<II", cid, len(e)) for cid, e in index.items())))
    index_pos = pos
    tail = b"".join(conn_recs[c] for c in sorted(conn_recs)) + b"".join(chunk_infos)
    hdr_fields = _fields({"op": bytes([OP_BAG_HEADER]), "index_pos": struct.pack("<Q", index_pos), "conn_count": struct.pack("<I", len(conn_recs)),
                          "chunk_count": struct.pack("<I", len(chunks))})
    pad = 4096 - 4 - len(hdr_fields) - 4
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(hdr_fields)) + hdr_fields + struct.pack("<I", pad) + b" " * pad)
        f.write(b"".join(body))
        f.write(tail)


def read_person2d_stream(path, topics=None):
    """-> {topic: [harness.CameraMsg]} for every person_msgs/Person2DList connection (or the given topics), in bag order."""
    from .harness import CameraMsg
    conns, msgs = read_bag(path)
    want = set(topics) if topics else {c["topic"] for c in conns.values() if c["type"] == "person_msgs/Person2DList"}
    out = {t: [] for t in sorted(want)}
    for topic, _, payload in msgs:
        if topic in want:
            m = deserialize_person2d_list(payload)
            out[topic].append(CameraMsg(m["stamp_ns"] * 1e-9, m["fb_delay"], m["persons"]))
    return out
