"""MI355X-native visual-MPC CEM planner (hot path of SudeepDasari/visual_foresight)."""
__version__ = '0.1.0'
