"""CPU oracle for the Transporter / KeyNet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``keypoints_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.
"""
