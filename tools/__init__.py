"""Measurement and evidence tooling (not product code)."""
