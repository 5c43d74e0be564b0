"""oracle/metrics_oracle.py against the episode sums captured from the reference env with the driver's
own bookkeeping statements (tools/capture_golden_metrics.py), and the host side of the scalar sink
(TFRecord framing, CRC32C, protobuf wire format).  CPU only."""
import glob
import os
import struct

import numpy as np
import pytest

from oracle import metrics_oracle as MO

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PATHS = sorted(glob.glob(os.path.join(GOLD, "episode_metrics_*.npz")))


@pytest.mark.parametrize("path", PATHS, ids=os.path.basename)
def test_episode_sums(path):
    d = np.load(path)
    V, n_env, n_ep, n_step = (int(x) for x in d["shape"])
    clipped = 0
    for e in range(n_env):
        orc = MO.EpisodeOracle(V)
        for ep in range(n_ep):
            orc.begin_episode()
            for st in range(n_step):
                orc.accumulate(d["metrics"][e, ep, st], d["reward"][e, ep, st], d["power_w"][e, ep, st])
            got = np.array([orc.end_episode()[c] for c in MO.COLUMNS])
            np.testing.assert_allclose(got, d["episode"][e, ep], rtol=1e-13, atol=1e-15, err_msg="env %d ep %d" % (e, ep))
            clipped += int((np.abs(d["reward"][e, ep]) > 5).sum())
    if V == 8:
        assert clipped > 0, "the fixture must exercise the +-5 clip of TRAIN:1714"


@pytest.mark.parametrize("path", PATHS, ids=os.path.basename)
def test_jain_index(path):
    d = np.load(path)
    for x, y in zip(d["jain_x"], d["jain_y"]):
        assert MO.jain_index(x) == y
    assert MO.jain_index(np.zeros(0)) == 0.0


def test_columns_match_the_product_and_the_header():
    from ris_vec_marl_amd import metrics as M, _native as N
    assert M.COLUMNS == MO.COLUMNS and len(M.COLUMNS) == N.EP_COLS
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "risvec.h")).read()
    assert "#define RISVEC_EP_FIXED %d" % N.EP_FIXED in hdr and "#define RISVEC_EP_COLS %d" % N.EP_COLS in hdr
    for name in ("risvec_episode_clear", "risvec_episode_accumulate", "risvec_episode_partial_rows",
                 "risvec_episode_summary"):
        assert name in N.EXPORTS and name in hdr


def test_crc32c_known_answers():
    from ris_vec_marl_amd.metrics import crc32c
    assert crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value (RFC 3720 B.4 polynomial)
    assert crc32c(b"") == 0
    assert crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # 32 bytes of ones
    assert crc32c(bytes(range(32))) == 0x46DD794E        # 32 incrementing bytes


def test_event_file_round_trip(tmp_path):
    from ris_vec_marl_amd.metrics import ScalarSink, read_events
    with ScalarSink(str(tmp_path)) as sink:
        sink.add_scalars({"delay/episode_mean": 0.25, "reward/jain": 0.875}, 3, walltime=12.5)
        sink.add_scalar("queue/mec_cycles", 6.5e7, 4, walltime=13.0)
        sink.add_scalar("neg/step", -1.5, -2, walltime=14.0)
    assert os.path.basename(sink.path).startswith("events.out.tfevents.")
    assert read_events(sink.path) == [(12.5, 3, "delay/episode_mean", 0.25), (12.5, 3, "reward/jain", 0.875),
                                      (13.0, 4, "queue/mec_cycles", float(np.float32(6.5e7))), (14.0, -2, "neg/step", -1.5)]
    lines = open(os.path.join(str(tmp_path), "scalars.jsonl")).read().splitlines()
    assert len(lines) == 3 and '"reward/jain": 0.875' in lines[0]
    blob = bytearray(open(sink.path, "rb").read())
    blob[-6] ^= 1                                          # a flipped payload bit must be caught by the CRC
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(blob))
    with pytest.raises(ValueError):
        read_events(str(bad))


def test_event_payload_parses_as_protobuf():
    """The payload bytes against an independent decoder: google.protobuf with Event / Summary
    descriptors built at run time (field numbers of tensorflow's event.proto / summary.proto)."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from ris_vec_marl_amd.metrics import encode_event, frame_record
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="risvec_event_test.proto", package="rvt", syntax="proto3")
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    val.field.add(name="simple_value", number=2, type=F.TYPE_FLOAT, label=F.LABEL_OPTIONAL)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=F.TYPE_MESSAGE, type_name=".rvt.Value", label=F.LABEL_REPEATED)
    ev = fd.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    ev.field.add(name="step", number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ev.field.add(name="file_version", number=3, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    ev.field.add(name="summary", number=5, type=F.TYPE_MESSAGE, type_name=".rvt.Summary", label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("rvt.Event")
    Event = message_factory.GetMessageClass(desc) if hasattr(message_factory, "GetMessageClass") \
        else message_factory.MessageFactory(pool).GetPrototype(desc)
    m = Event()
    m.ParseFromString(encode_event(1.5e9, 123456789012, [("a/b", 0.5), ("c", -3.0)]))
    assert m.wall_time == 1.5e9 and m.step == 123456789012
    assert [(v.tag, v.simple_value) for v in m.summary.value] == [("a/b", 0.5), ("c", -3.0)]
    m = Event()
    m.ParseFromString(encode_event(2.0, -7, file_version="brain.Event:2"))
    assert m.step == -7 and m.file_version == "brain.Event:2" and len(m.summary.value) == 0
    # and the other way round: what protobuf serialises is what the sink wrote
    m = Event(wall_time=3.25, step=9)
    v = m.summary.value.add()
    v.tag, v.simple_value = "x", 1.25
    assert m.SerializeToString() == encode_event(3.25, 9, [("x", 1.25)])
    rec = frame_record(b"abc")
    assert struct.unpack("<Q", rec[:8])[0] == 3 and len(rec) == 8 + 4 + 3 + 4
