#!/usr/bin/env python3
"""Entry point in the reference's place (main.py:441-448 starts App(800, 800)): opens the
blit-only window when glfw/PyOpenGL and a display exist, otherwise renders headless to PNG.

    python main.py [--width W --height H] [--headless --frames N --png frame_%04d.png]
"""
import rt355

if __name__ == "__main__":
    rt355.app.main()
