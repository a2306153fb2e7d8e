"""The image-generation workers' device map (no GPU needed): worker t of generate_images (variant) and of make_images (polish) works
on device_ids[t % n], as the reference's run_inference deals its callers over device_ids (RunInference.py:101-116)."""
from types import SimpleNamespace


def test_polish_worker_device_map():
    from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport, parse_device_ids
    assert parse_device_ids("0,2,5") == [0, 2, 5] and parse_device_ids(" 3 ") == [3]
    assert parse_device_ids(None) == [0] and parse_device_ids("") == [0] and parse_device_ids([1, 4]) == [1, 4] and parse_device_ids(6) == [6]
    assert [UserInterfaceSupport.worker_device("0,1,2", t) for t in range(7)] == [0, 1, 2, 0, 1, 2, 0]
    assert [UserInterfaceSupport.worker_device(None, t) for t in range(3)] == [0, 0, 0]


def test_variant_worker_device_map():
    from pepper_amd.variant.ImageGenerationUI import _on_device, worker_device
    assert [worker_device(SimpleNamespace(device_ids="0,1,2,3"), t) for t in range(6)] == [0, 1, 2, 3, 0, 1]
    assert [worker_device(SimpleNamespace(device_ids=[4, 6]), t) for t in range(3)] == [4, 6, 4]
    # image_device_ids wins over the inference step's device_ids; neither: options.device, default 0
    assert worker_device(SimpleNamespace(device_ids="0,1", image_device_ids="7"), 5) == 7
    assert worker_device(SimpleNamespace(device=3), 9) == 3 and worker_device(SimpleNamespace(), 9) == 0
    o = SimpleNamespace(device=0, bam="x")
    assert _on_device(o, 0) is o
    clone = _on_device(o, 2)
    assert clone.device == 2 and clone.bam == "x" and o.device == 0
