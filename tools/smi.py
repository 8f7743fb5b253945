"""Measurement helper (bench.py's `energy` object, tools/energy_probe.py): the package energy accumulator, power, clocks and
the power cap of one GPU through librocm_smi64 (ctypes; no rocm-smi subprocess in a timed region).  Not product code.

The energy accumulator (rsmi_dev_energy_count_get) counts in units of `resolution` micro-joules (15.26 uJ on MI300-class parts)
and is updated by the SMU about once per millisecond: differences over regions of >= 100 ms are good to < 1 %.
"""
import ctypes as C
import ctypes.util
import os


class Smi:
    def __init__(self, torch_device=None):
        path = None
        for cand in ("/opt/rocm/lib/librocm_smi64.so", ctypes.util.find_library("rocm_smi64")):
            if cand and (os.path.exists(cand) or "/" not in cand):
                path = cand
                break
        if path is None:
            raise OSError("librocm_smi64 not found")
        self.lib = C.CDLL(path)
        rc = self.lib.rsmi_init(C.c_uint64(0))
        if rc != 0:
            raise OSError(f"rsmi_init failed ({rc})")
        n = C.c_uint32(0)
        self.lib.rsmi_num_monitor_devices(C.byref(n))
        self.n = int(n.value)
        self.idx = self._match(torch_device) if torch_device is not None else 0

    def _match(self, dev):
        """rsmi index of a torch device, by PCI bus / device number (HIP's order follows HIP_VISIBLE_DEVICES, rsmi's does not)"""
        import torch

        p = torch.cuda.get_device_properties(dev)
        want = (getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None))
        if want[0] is not None:
            for i in range(self.n):
                bdf = C.c_uint64(0)
                if self.lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) == 0:
                    if ((bdf.value >> 8) & 0xff, (bdf.value >> 3) & 0x1f) == want:
                        return i
        idx = torch.device(dev).index or 0
        return idx if idx < self.n else 0

    # ---- readings ---------------------------------------------------------------------------------------------------
    def energy_uj(self):
        """(micro-joules since an arbitrary origin, timestamp ns) -- raises OSError when the accumulator is not readable"""
        e, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
        rc = self.lib.rsmi_dev_energy_count_get(C.c_uint32(self.idx), C.byref(e), C.byref(res), C.byref(ts))
        if rc != 0:
            raise OSError(f"rsmi_dev_energy_count_get failed ({rc})")
        self.resolution_uj = float(res.value)
        return e.value * float(res.value), int(ts.value)

    def power_w(self):
        p = C.c_uint64(0)
        t = C.c_int(0)
        if self.lib.rsmi_dev_power_get(C.c_uint32(self.idx), C.byref(p), C.byref(t)) == 0:
            return p.value / 1e6
        if self.lib.rsmi_dev_current_socket_power_get(C.c_uint32(self.idx), C.byref(p)) == 0:
            return p.value / 1e6
        return None

    def sclk_mhz(self):
        class Freqs(C.Structure):
            _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]
        f = Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.idx), C.c_int(0), C.byref(f)) != 0 or f.current >= 33:   # RSMI_CLK_TYPE_SYS
            return None
        return f.frequency[f.current] / 1e6

    def cap_w(self):
        c = C.c_uint64(0)
        rc = self.lib.rsmi_dev_power_cap_get(C.c_uint32(self.idx), C.c_uint32(0), C.byref(c))
        return c.value / 1e6 if rc == 0 else None

    def cap_range_w(self):
        hi, lo = C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.rsmi_dev_power_cap_range_get(C.c_uint32(self.idx), C.c_uint32(0), C.byref(hi), C.byref(lo))
        return (lo.value / 1e6, hi.value / 1e6) if rc == 0 else None


class EnergyMeter:
    """joules between start() and stop(); `ok` False (and a reason) where the accumulator cannot be read"""

    def __init__(self, torch_device=None):
        self.ok, self.why = True, None
        try:
            self.smi = Smi(torch_device)
            self.smi.energy_uj()
        except Exception as e:  # noqa: BLE001
            self.ok, self.why, self.smi = False, f"{type(e).__name__}: {e}", None

    def start(self):
        if self.ok:
            self.e0, self.t0 = self.smi.energy_uj()

    def stop(self):
        """-> (joules, seconds by the counter's own timestamps) or None"""
        if not self.ok:
            return None
        e1, t1 = self.smi.energy_uj()
        return (e1 - self.e0) * 1e-6, (t1 - self.t0) * 1e-9
