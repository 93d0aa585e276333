"""rc_mvsnet_amd: MI355X-native plane-sweep hot path of RC-MVSNet (import as ``rc_mvsnet_amd``)."""
