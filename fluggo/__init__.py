"""Namespace of the Fluggo media library; this repository provides `fluggo.media.process` (MI355X build)
and `fluggo.media.basetypes`."""
