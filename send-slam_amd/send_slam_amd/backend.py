"""HipBackend -- backend-lifecycle mirror of SendSlam.DockerHandler for the HIP front door.

Same call contract and reply shapes as send_slam/lib/send_slam/docker_handler.ex:15-21,
:40-43, :82-115, with `docker run / inspect / logs / rm -f` replaced by a child process
running send-slam_amd/frontdoor/sendslam_frontdoor on one GPU:

    start_container() -> ("ok", id) | ("error", reason)        docker run -d      :154-165
    stop_container()  -> "ok" | ("error", reason)              docker rm -f       :167-175
    status()          -> {state, container_id, last_seen}      :106-109
    logs(lines=100)   -> ("ok", text)                          docker logs --tail :177-182

Environment merging follows :195-205: `ORBSLAM3_`-prefixed OS variables (prefix stripped),
then the runtime `env` option; the listener port reaches the child as ORB_SLAM3_WS_PORT
(orbslam3_mono_networked.cc:346).  The Elixir original of this class is
send-slam_amd/nif/hip_backend.ex.
"""
from __future__ import annotations

import collections
import os
import subprocess
import threading
import time
from typing import Dict, Optional

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONTDOOR = os.path.join(_PKG, "frontdoor", "sendslam_frontdoor")


def os_prefixed_env(prefix: str = "ORBSLAM3_") -> Dict[str, str]:
    return {k[len(prefix):]: v for k, v in os.environ.items() if k.startswith(prefix)}


class HipBackend:
    def __init__(self, port: int = 5000, device: int = 0, name: str = "net-orbslam", env: Optional[dict] = None,
                 binary: str = FRONTDOOR, log_lines: int = 2000, auto_restart: bool = False):
        merged = dict(os_prefixed_env())
        merged.update({str(k): str(v) for k, v in (env or {}).items()})
        merged.setdefault("ORB_SLAM3_WS_PORT", str(port))
        merged.setdefault("SENDSLAM_DEVICE", str(device))
        self.env_map = merged
        self.name = name
        self.binary = binary
        self.state = "initial"
        self.container_id: Optional[str] = None
        self.last_seen: Optional[int] = None
        self._proc: Optional[subprocess.Popen] = None
        self._log = collections.deque(maxlen=log_lines)
        self._reader: Optional[threading.Thread] = None
        # auto_restart (application.ex:94 passes it, DockerHandler never reads it): poll() relaunches an
        # exited backend instead of only reporting it -- the reference's restart path never relaunches
        # (application.ex:117 is the single start_container call)
        self.auto_restart = auto_restart
        self.restarts = 0

    def start_container(self):
        if self.container_id is not None:
            return ("ok", self.container_id)
        if not os.path.exists(self.binary):
            self.state = "error"
            return ("error", f"{self.binary} is not built (make -C send-slam_amd/frontdoor)")
        env = dict(os.environ)
        env.update(self.env_map)
        try:
            self._proc = subprocess.Popen([self.binary], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                          text=True, bufsize=1)
        except OSError as e:
            self.state = "error"
            return ("error", str(e))
        self._reader = threading.Thread(target=self._pump, daemon=True)
        self._reader.start()
        self.container_id = f"{self.name}-{self._proc.pid}"
        self.state = "running"
        self.last_seen = int(time.monotonic() * 1000)
        return ("ok", self.container_id)

    def _pump(self):
        assert self._proc and self._proc.stdout
        for line in self._proc.stdout:
            self._log.append(line.rstrip("\n"))

    def poll(self):
        """The :poll message: running -> refresh last_seen; exited -> state :exited."""
        if self._proc is None:
            return ("error", "no_container")
        if self._proc.poll() is None:
            self.state = "running"
            self.last_seen = int(time.monotonic() * 1000)
            return ("ok", self.state)
        self.state = "exited"
        if self.auto_restart:
            self.container_id = None
            self.restarts += 1
            return self.start_container()
        return ("error", "container_not_running")

    def stop_container(self):
        if self._proc is not None and self._proc.poll() is None:
            self._proc.terminate()
            try:
                self._proc.wait(timeout=15)
            except subprocess.TimeoutExpired:
                self._proc.kill()
                self._proc.wait()
        self.state = "exited"
        return "ok"

    def wait(self, timeout: Optional[float] = None) -> Optional[int]:
        if self._proc is None:
            return None
        rc = self._proc.wait(timeout=timeout)
        if self._reader:
            self._reader.join(timeout=2)
        self.state = "exited"
        return rc

    def status(self):
        return {"state": self.state, "container_id": self.container_id, "last_seen": self.last_seen}

    def logs(self, lines: int = 100):
        return ("ok", "\n".join(list(self._log)[-lines:]))
