"""send_slam_amd -- host-side mirror of SEND-SLAM's backend interface over libsendslam_orb.so.

Only what the ORB extract + match path needs: the ctypes binding of the C ABI
(`binding`), seeded synthetic frames (`synth`), the wire protocol of the reference's TCP
link (`wire`) and the backend-lifecycle mirror (`backend`).
"""
__all__ = ["binding", "synth", "wire", "backend"]
