/* placeholder, replaced below */
