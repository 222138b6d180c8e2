"""The helpers of /root/reference/hive/utils.py that the hot path uses."""
import contextlib
import datetime
import logging
from typing import Optional

import numpy as np


def usable_cores() -> int:
    """Cores this process may really use: the scheduler affinity, cut down to the cgroup's CPU quota where one is set (a container on a 256-thread host is often given a
    share of 8-16: sizing a thread pool by os.cpu_count() there makes it slower, not faster)."""
    import os
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def num2str(num: Optional[int]):
    return '?' if num is None else str(num)


def validate_shape(x: np.ndarray, x_name: str, expected_shape: tuple):
    """Assert that ``x.shape`` matches ``expected_shape`` (``None`` = any size), with the
    reference's AssertionError messages (/root/reference/hive/utils.py:38-63)."""
    assert type(expected_shape) is tuple, "`expected_shape` must be a tuple."
    assert len(x.shape) == len(expected_shape), \
        f"Incorrect number of dimensions for {x_name}; expected {len(expected_shape)} but got {len(x.shape)}"
    ok = all(e is None or s == e for s, e in zip(x.shape, expected_shape))
    assert ok, f"Incorrect shape for {x_name}: expected ({', '.join(map(num2str, expected_shape))}) but got {x.shape}"


def validate_camera_parameter_shapes(K, R, t):
    validate_shape(K, 'K', expected_shape=(3, 3))
    validate_shape(R, 'R', expected_shape=(3, 3))
    validate_shape(t, 't', expected_shape=(3, 1))


class Timer:
    """Wall-clock timer (utils.py:253-299)."""

    def __init__(self):
        self.start_time = None
        self.stop_time = None

    def start(self):
        self.start_time = datetime.datetime.now()
        return self

    def stop(self):
        self.stop_time = datetime.datetime.now()
        return self

    @property
    def elapsed(self) -> datetime.timedelta:
        return (self.stop_time or datetime.datetime.now()) - self.start_time


def set_key_path(d: dict, key_path, value):
    for key in key_path[:-1]:
        d = d.setdefault(key, {})
    d[key_path[-1]] = value


@contextlib.contextmanager
def timed_block(log_msg: str, profiling: Optional[dict] = None, key_path=None):
    """Time a block, log it, and record the elapsed seconds at ``key_path`` of ``profiling``
    (utils.py:356-379)."""
    timer = Timer().start()
    try:
        yield timer
    finally:
        timer.stop()
        logging.info(f"{log_msg} {timer.elapsed}")
        if profiling is not None and key_path:
            set_key_path(profiling, list(key_path), timer.elapsed.total_seconds())
