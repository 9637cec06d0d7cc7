"""Host-side lowering of the UNet to HIP kernel launches."""
