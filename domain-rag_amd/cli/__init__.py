"""Stage CLIs with the reference's flags and on-disk formats (domainrag.sh stage boundaries)."""
