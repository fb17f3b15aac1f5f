"""MI355X-native ScaNN hot path (gfx950).  See DESIGN.md."""
